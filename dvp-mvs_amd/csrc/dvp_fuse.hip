// dvp_fuse.hip — depth-map fusion on the GPU: the reference's RunFusion (ETH3D variant, /root/reference/APD.cpp:1809-1960)
// behind the dvp_fuse_* entry points of include/dvp_mvs.h.
//
// What the reference computes, view after view in pair.txt order, pixel after pixel in raster order: a pixel with a positive
// depth that no earlier point has claimed is lifted to the world point X; X is dropped into every source view; the pixel it
// lands on is a WITNESS if it is unclaimed, has a depth and agrees with the reference pixel three ways (reprojection error
// < 2 px, relative depth difference < 1 %, normals within 10 degrees); each witness votes exp(-(e + 200 r + 10 a)); the point
// is kept when it has a witness and the mean vote exceeds 0.3 (0.45 for a WEAK pixel); a kept point claims its witnesses.
//
// The claims are what makes the scan sequential — and only they.  Here, per reference view:
//   1. fuse_gather   one thread per pixel: the candidate witnesses (source pixel + vote) of the pixel.  Nothing in it depends
//                    on the scan order: witnesses claimed by EARLIER VIEWS are final and dropped here, witnesses claimed by
//                    earlier pixels of THIS view are the next step's question.
//   2. resolve       claimed_at(p, w) = "some accepted pixel q < p of this view lists w" (an accepted q leaves w claimed whether
//                    it was q's witness or somebody else's before).  So a pixel can be decided as soon as every earlier
//                    pixel that lists one of its candidates has been decided.  Rounds of two launches over the undecided
//                    pixels: (a) every undecided pixel writes its index to its candidates with atomicMin; (b) a pixel
//                    that reads back its own index everywhere is the earliest undecided lister of all its candidates: it
//                    decides — live witnesses = candidates without an accepted earlier lister, votes summed in source order
//                    exactly as the sequential scan does — and, if accepted, leaves its index at its witnesses (atomicMin
//                    again: the EARLIEST accepted lister is what later pixels compare with).  Conflicts are local (two
//                    pixels share a witness only when they project to the same source pixel), so a few rounds decide
//                    almost everything; the tail of a long chain is finished in index order by one lane.
//                    Both per-source-pixel words carry a generation in their high half, so nothing is ever reset.
//   3. fuse_emit     accepted pixels in raster order (block counts + ranks inside a block): position, mean colour of the
//                    pixel and its witnesses, and the witnesses' claim flags.
// Same points, same order, same bits as the sequential scan (tests/test_host_oracles.py).
#include "dvp_fuse_math.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace dvp;

namespace {

struct FuseView {
	DvpCamera cam;
	float centre[3];            // -R^T t in binary32 (APD.cpp:515-518 recomputes it; Camera::c is accumulated in double)
	int cols, rows;
	const float* depth;         // [rows * cols]
	const float* normal;        // [rows * cols * 3]
	const uint8_t* weak;        // [rows * cols] or null (all STRONG)
	const uint8_t* bgr;         // [rows * cols * 3]
	const uint8_t* block;       // [rows * cols] or null: reference pixels below 128 are excluded (APD.cpp:1885-1887)
	uint8_t* claimed;           // [rows * cols]
	unsigned long long* und;    // [rows * cols] (~round) << 32 | earliest undecided lister of this pixel in the current round
	unsigned long long* acc;    // [rows * cols] (~view serial) << 32 | earliest accepted lister of this pixel in the current view
};

// pixel + depth -> world (APD.cpp:502-523)
__device__ __forceinline__ f3 fuse_lift(const FuseView& v, int x, int y, float z) {
	const float cx = z * (x - v.cam.K[2]) / v.cam.K[0];
	const float cy = z * (y - v.cam.K[5]) / v.cam.K[4];
	f3 w;
	w.x = (v.cam.R[0] * cx + v.cam.R[3] * cy + v.cam.R[6] * z) + v.centre[0];
	w.y = (v.cam.R[1] * cx + v.cam.R[4] * cy + v.cam.R[7] * z) + v.centre[1];
	w.z = (v.cam.R[2] * cx + v.cam.R[5] * cy + v.cam.R[8] * z) + v.centre[2];
	return w;
}
// ProjectCamera (APD.cpp:536-546)
__device__ __forceinline__ void fuse_project(const f3 X, const DvpCamera& cam, f2* pt, float* depth) {
	f3 t;
	t.x = cam.R[0] * X.x + cam.R[1] * X.y + cam.R[2] * X.z + cam.t[0];
	t.y = cam.R[3] * X.x + cam.R[4] * X.y + cam.R[5] * X.z + cam.t[1];
	t.z = cam.R[6] * X.x + cam.R[7] * X.y + cam.R[8] * X.z + cam.t[2];
	*depth = cam.K[6] * t.x + cam.K[7] * t.y + cam.K[8] * t.z;
	pt->x = (cam.K[0] * t.x + cam.K[1] * t.y + cam.K[2] * t.z) / *depth;
	pt->y = (cam.K[3] * t.x + cam.K[4] * t.y + cam.K[5] * t.z) / *depth;
}
// int(v + 0.5f) as the host's conversion gives it: a value no int holds (NaN, +-huge) lands outside every image
__device__ __forceinline__ int fuse_round(float v) {
	const float t = v + 0.5f;
	if (!(t > -2147483648.0f && t < 2147483648.0f)) return -1;
	return (int)t;
}

struct GatherArgs {
	const FuseView* views;
	int ref, ns;
	const int* src;             // [ns] view slots in pair.txt order
	int* cand_view;             // [ns][L]  slot of the source view (position in the source list is implied by the order kept)
	int* cand_pix;              // [ns][L]
	float* cand_vote;           // [ns][L]
	signed char* count;         // [L] candidates of the pixel, -1 = not a reference pixel
};

__global__ void __launch_bounds__(256) fuse_gather(const GatherArgs a) {
	const FuseView& R = a.views[a.ref];
	const int W = R.cols, H = R.rows;
	const size_t L = (size_t)W * H;
	const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (p >= L) return;
	const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
	int n = -1;
	const float z = R.depth[p];
	if (!(R.block && R.block[p] < 128) && !(R.claimed[p] == 1 || z <= 0.0f)) {
		n = 0;
		const f3 X = fuse_lift(R, x, y, z);
		const float nr[3] = { R.normal[3 * p], R.normal[3 * p + 1], R.normal[3 * p + 2] };
		for (int j = 0; j < a.ns; ++j) {
			const FuseView& S = a.views[a.src[j]];
			f2 q;
			float zq;
			fuse_project(X, S.cam, &q, &zq);
			const int sx = fuse_round(q.x), sy = fuse_round(q.y);
			if (sx < 0 || sx >= S.cols || sy < 0 || sy >= S.rows) continue;
			const size_t sp = (size_t)sy * S.cols + sx;
			const float zs = S.depth[sp];
			if (zs <= 0.0f) continue;
			if (S.claimed[sp] == 1) continue;   // claimed by an earlier view: final (claims of THIS view's scan are the resolve step's)
			f2 back;
			float z_seen;
			fuse_project(fuse_lift(S, sx, sy, zs), R.cam, &back, &z_seen);
			const double ex = (double)(x - back.x), ey = (double)(y - back.y);   // std::pow(float, int) is double (APD.cpp:1918)
			const float err = (float)sqrt(ex * ex + ey * ey);
			const float rel = fabsf(z_seen - z) / z;
			const float ns3[3] = { S.normal[3 * sp], S.normal[3 * sp + 1], S.normal[3 * sp + 2] };
			const float ang = fuse_angle(nr, ns3);
			if (err < 2.0f && rel < 0.01f && ang < 0.174533f) {
				a.cand_view[(size_t)n * L + p] = a.src[j];
				a.cand_pix[(size_t)n * L + p] = (int)sp;
				a.cand_vote[(size_t)n * L + p] = fuse_expf(-(err + 200 * rel + ang * 10));
				++n;
			}
		}
	}
	a.count[p] = (signed char)n;
}

// pixels with at least one candidate -> the first undecided list (any order)
__global__ void __launch_bounds__(256) fuse_first_list(const signed char* count, size_t L, unsigned* list, unsigned* n_list) {
	const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
	const bool go = p < L && count[p] > 0;
	const unsigned long long m = __ballot(go);
	if (!m) return;
	const int lane = threadIdx.x & 63;
	unsigned base = 0;
	if (lane == __builtin_ctzll(m)) base = atomicAdd(n_list, (unsigned)__popcll(m));
	base = __shfl(base, __builtin_ctzll(m), 64);
	if (go) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)p;
}

struct ResolveArgs {
	const FuseView* views;
	int ref, ns;
	const int* cand_view;
	const int* cand_pix;
	const float* cand_vote;
	const signed char* count;
	uint8_t* decision;          // [L] 0 = undecided / not a reference pixel, 1 = accepted, 2 = rejected
	unsigned long long* live;   // [L] bit k: candidate k was a witness when the pixel was decided
	size_t L;
	unsigned round_hi, view_hi; // ~round, ~view serial: the generation halves of FuseView::und / ::acc
	const unsigned* list;       // undecided pixels of this round
	unsigned n_list;
	unsigned* next;             // ... of the next one
	unsigned* n_next;
};

__global__ void __launch_bounds__(256) fuse_mark(const ResolveArgs a) {
	const unsigned i = blockIdx.x * 256 + threadIdx.x;
	if (i >= a.n_list) return;
	const unsigned p = a.list[i];
	const int n = a.count[p];
	const unsigned long long key = ((unsigned long long)a.round_hi << 32) | p;
	for (int k = 0; k < n; ++k)
		atomicMin(&a.views[a.cand_view[(size_t)k * a.L + p]].und[a.cand_pix[(size_t)k * a.L + p]], key);
}

// the decision of pixel p given that every earlier lister of its candidates has been decided
__device__ __forceinline__ void fuse_decide_px(const ResolveArgs& a, unsigned p) {
	const FuseView& R = a.views[a.ref];
	const int n = a.count[p];
	float votes = 0.0f;
	int nw = 0;
	unsigned long long live = 0;
	for (int k = 0; k < n; ++k) {
		const unsigned long long e = a.views[a.cand_view[(size_t)k * a.L + p]].acc[a.cand_pix[(size_t)k * a.L + p]];
		if ((unsigned)(e >> 32) == a.view_hi && (unsigned)e < p) continue;   // claimed by an accepted earlier pixel of this view
		live |= 1ull << k;
		votes += a.cand_vote[(size_t)k * a.L + p];
		++nw;
	}
	const float needed = (R.weak && R.weak[p] == DVP_WEAK) ? 0.45f : 0.3f;
	const bool accept = nw >= 1 && votes > needed * nw;
	a.live[p] = live;
	if (accept) {
		const unsigned long long key = ((unsigned long long)a.view_hi << 32) | p;
		for (int k = 0; k < n; ++k)
			if ((live >> k) & 1) atomicMin(&a.views[a.cand_view[(size_t)k * a.L + p]].acc[a.cand_pix[(size_t)k * a.L + p]], key);
	}
	a.decision[p] = accept ? 1 : 2;
}

__global__ void __launch_bounds__(256) fuse_decide(const ResolveArgs a) {
	const unsigned i = blockIdx.x * 256 + threadIdx.x;
	bool later = false;
	unsigned p = 0;
	if (i < a.n_list) {
		p = a.list[i];
		const int n = a.count[p];
		const unsigned long long key = ((unsigned long long)a.round_hi << 32) | p;
		bool first = true;
		for (int k = 0; k < n; ++k)
			first = first && a.views[a.cand_view[(size_t)k * a.L + p]].und[a.cand_pix[(size_t)k * a.L + p]] == key;
		if (first) fuse_decide_px(a, p);
		else later = true;
	}
	const unsigned long long m = __ballot(later);
	if (!m) return;
	const int lane = threadIdx.x & 63;
	unsigned base = 0;
	if (lane == __builtin_ctzll(m)) base = atomicAdd(a.n_next, (unsigned)__popcll(m));
	base = __shfl(base, __builtin_ctzll(m), 64);
	if (later) a.next[base + __popcll(m & ((1ull << lane) - 1ull))] = p;
}

// what the rounds left over, in index order (the list is sorted by the host), one lane: the sequential scan itself
__global__ void fuse_decide_rest(const ResolveArgs a) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	for (unsigned i = 0; i < a.n_list; ++i) {
		fuse_decide_px(a, a.list[i]);
		__threadfence();
	}
}

struct EmitArgs {
	const FuseView* views;
	int ref;
	const int* cand_view;
	const int* cand_pix;
	const signed char* count;
	const uint8_t* decision;
	const unsigned long long* live;
	size_t L;
	unsigned* block_count;      // accepted pixels per block of 1024
	const unsigned long long* block_base;   // exclusive scan of block_count
	float* out;                 // [points][6]: x y z b g r
};

__global__ void __launch_bounds__(1024) fuse_count(const EmitArgs a) {
	__shared__ unsigned wave_n[16];
	const size_t p = (size_t)blockIdx.x * 1024 + threadIdx.x;
	const bool acc = p < a.L && a.decision[p] == 1;
	const unsigned long long m = __ballot(acc);
	if ((threadIdx.x & 63) == 0) wave_n[threadIdx.x >> 6] = (unsigned)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned s = 0;
		for (int w = 0; w < 16; ++w) s += wave_n[w];
		a.block_count[blockIdx.x] = s;
	}
}

__global__ void __launch_bounds__(1024) fuse_emit(const EmitArgs a) {
	__shared__ unsigned wave_n[16];
	const FuseView& R = a.views[a.ref];
	const size_t p = (size_t)blockIdx.x * 1024 + threadIdx.x;
	const bool acc = p < a.L && a.decision[p] == 1;
	const unsigned long long m = __ballot(acc);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_n[wave] = (unsigned)__popcll(m);
	__syncthreads();
	if (!acc) return;
	unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
	for (int w = 0; w < wave; ++w) rank += wave_n[w];
	const int y = (int)(p / R.cols), x = (int)(p - (size_t)y * R.cols);
	const f3 X = fuse_lift(R, x, y, R.depth[p]);
	// the pixel's own colour, then the witnesses' in source order (APD.cpp:1935-1946): small integers, every partial sum exact
	float sb = 0.0f, sg = 0.0f, sr = 0.0f;
	int nw = 0;
	const unsigned long long live = a.live[p];
	const int n = a.count[p];
	for (int k = 0; k < n; ++k) {
		if (!((live >> k) & 1)) continue;
		const FuseView& S = a.views[a.cand_view[(size_t)k * a.L + p]];
		const size_t sp = (size_t)a.cand_pix[(size_t)k * a.L + p];
		sb += S.bgr[3 * sp]; sg += S.bgr[3 * sp + 1]; sr += S.bgr[3 * sp + 2];
		S.claimed[sp] = 1;
		++nw;
	}
	float* o = a.out + (a.block_base[blockIdx.x] + rank) * 6;
	o[0] = X.x; o[1] = X.y; o[2] = X.z;
	o[3] = ((float)R.bgr[3 * p] + sb) / (nw + 1);
	o[4] = ((float)R.bgr[3 * p + 1] + sg) / (nw + 1);
	o[5] = ((float)R.bgr[3 * p + 2] + sr) / (nw + 1);
}

// ---- the Tanks & Temples variants (RunFusion_TAT_Intermediate / _advanced, APD.cpp:1962-2130 / 2132-2279) ------------------
// Same geometry per (pixel, source); what differs: a pixel is NOT skipped for being claimed, witnesses that are claimed do not
// count, a kept pixel claims ITSELF — so the claim flags a view reads belong to OTHER views and stand still during its scan —
// and the acceptance is graded: the first k = 2 .. #sources with at least k sources inside k-scaled thresholds.  The one
// sequential thing is a quirk the port keeps: the per-source residuals are ONE array per view, overwritten only when a source
// yields a comparison — a source that drops out keeps voting with the residuals of the last pixel it was compared for
// (APD.cpp:2051, 2078-2090).  "The last pixel q <= p with a comparison for source j" is an inclusive maximum scan of
// (compared ? q : -1) over the raster order:
//   graded_gather   per pixel and source: the comparison (error, depth difference, angle, source pixel) or none; per block
//                   and source the last pixel with a comparison
//   (host)          running maximum over the blocks -> what each block inherits
//   graded_decide   per block: the scan inside the block, then per pixel the residuals in force, the smallest k each source
//                   agrees from (the thresholds grow with k), the first k with k agreeing sources, the agreeing set
//   graded_emit     accepted pixels in raster order, colours, self-claims
struct GradedArgs {
	const FuseView* views;
	int ref, ns, advanced;
	const int* src;             // [ns] view slots, -1 = a source without maps (it never yields a comparison)
	float* f_err; float* f_rel; float* f_ang; int* f_pix;   // [ns][L] comparisons, f_pix < 0: none
	uint8_t* is_ref;            // [L]
	int* block_last;            // [blocks][ns] last pixel of the block with a comparison for the source, -1
	const int* block_carry;     // [blocks][ns] ... of all earlier blocks
	int* last_idx;              // [ns][L] the pixel whose comparison is in force
	uint8_t* decision;          // [L] 1 = kept
	unsigned long long* agree;  // [L] the sources that agree at the accepted k
	size_t L;
};
constexpr int kGradedBlock = 256;

__global__ void __launch_bounds__(kGradedBlock) graded_gather(const GradedArgs a) {
	__shared__ int s_last[64];
	const FuseView& R = a.views[a.ref];
	const size_t L = a.L;
	const size_t p = (size_t)blockIdx.x * kGradedBlock + threadIdx.x;
	if (threadIdx.x < 64) s_last[threadIdx.x] = -1;
	__syncthreads();
	bool ref = false;
	if (p < L) {
		const int y = (int)(p / R.cols), x = (int)(p - (size_t)y * R.cols);
		const float z = R.depth[p];
		ref = !(R.block && R.block[p] < 128) && !(z <= 0.0f);
		if (ref) {
			const f3 X = fuse_lift(R, x, y, z);
			const float nr[3] = { R.normal[3 * p], R.normal[3 * p + 1], R.normal[3 * p + 2] };
			for (int j = 0; j < a.ns; ++j) {
				int pix = -1;
				float err = 0.0f, rel = 0.0f, ang = 0.0f;
				if (a.src[j] >= 0) {
					const FuseView& S = a.views[a.src[j]];
					f2 q;
					float zq;
					fuse_project(X, S.cam, &q, &zq);
					const int sx = fuse_round(q.x), sy = fuse_round(q.y);
					if (sx >= 0 && sx < S.cols && sy >= 0 && sy < S.rows) {
						const size_t sp = (size_t)sy * S.cols + sx;
						const float zs = S.depth[sp];
						if (!(S.claimed[sp] == 1 || zs <= 0.0f)) {
							f2 back;
							float z_seen;
							fuse_project(fuse_lift(S, sx, sy, zs), R.cam, &back, &z_seen);
							const double ex = (double)(x - back.x), ey = (double)(y - back.y);
							err = (float)sqrt(ex * ex + ey * ey);
							rel = fabsf(z_seen - z) / z;
							const float ns3[3] = { S.normal[3 * sp], S.normal[3 * sp + 1], S.normal[3 * sp + 2] };
							ang = fuse_angle(nr, ns3);
							pix = (int)sp;
						}
					}
				}
				a.f_err[(size_t)j * L + p] = err; a.f_rel[(size_t)j * L + p] = rel; a.f_ang[(size_t)j * L + p] = ang; a.f_pix[(size_t)j * L + p] = pix;
				if (pix >= 0) atomicMax(&s_last[j], (int)p);
			}
		} else {
			for (int j = 0; j < a.ns; ++j) a.f_pix[(size_t)j * L + p] = -1;
		}
		a.is_ref[p] = ref ? 1 : 0;
	}
	__syncthreads();
	if (threadIdx.x < a.ns) a.block_last[(size_t)blockIdx.x * a.ns + threadIdx.x] = s_last[threadIdx.x];
}

__global__ void __launch_bounds__(kGradedBlock) graded_decide(const GradedArgs a) {
	__shared__ int s_wave[kGradedBlock / 64];
	__shared__ uint8_t s_k[64 * kGradedBlock];     // [source][thread]: the smallest k the source agrees from (255: never)
	const float dist_base = 0.25f, depth_base = a.advanced ? 1.0f / 3000.0f : 1.0f / 3500.0f;
	const float angle_base = 0.06981317007977318f, angle_grad = 0.05235987755982988f;   // 4 and 3 degrees
	const size_t L = a.L;
	const size_t p = (size_t)blockIdx.x * kGradedBlock + threadIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const bool in = p < L;
	const bool ref = in && a.is_ref[p] != 0;
	for (int j = 0; j < a.ns; ++j) {
		// inclusive maximum scan of (comparison ? p : -1) over the block, then what the block inherits
		int v = (in && a.f_pix[(size_t)j * L + p] >= 0) ? (int)p : -1;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int t = __shfl_up(v, o, 64);
			if (lane >= o && t > v) v = t;
		}
		__syncthreads();                       // (s_wave of the previous source has been read)
		if (lane == 63) s_wave[wave] = v;
		__syncthreads();
		for (int w = 0; w < wave; ++w) if (s_wave[w] > v) v = s_wave[w];
		const int carry = a.block_carry[(size_t)blockIdx.x * a.ns + j];
		if (carry > v) v = carry;
		uint8_t kj = 255;
		if (ref) {
			a.last_idx[(size_t)j * L + p] = v;
			if (v >= 0) {
				const float err = a.f_err[(size_t)j * L + v], rel = a.f_rel[(size_t)j * L + v], ang = a.f_ang[(size_t)j * L + v];
				for (int k = 2; k <= a.ns; ++k)
					if (err < k * dist_base && rel < k * depth_base && (a.advanced || ang < (k * angle_grad + angle_base))) { kj = (uint8_t)k; break; }
			}
		}
		s_k[j * kGradedBlock + threadIdx.x] = kj;
	}
	if (!ref) { if (in) a.decision[p] = 0; return; }
	uint8_t dec = 0;
	unsigned long long agree = 0;
	for (int k = 2; k <= a.ns; ++k) {
		int count = 0;
		unsigned long long m = 0;
		for (int j = 0; j < a.ns; ++j)
			if (s_k[j * kGradedBlock + threadIdx.x] <= k) { ++count; m |= 1ull << j; }
		if (count >= k) { dec = 1; agree = m; break; }
	}
	a.decision[p] = dec;
	a.agree[p] = agree;
}

struct GradedEmitArgs {
	const FuseView* views;
	int ref, ns, advanced;
	const int* src;
	const int* last_idx;
	const int* f_pix;
	const uint8_t* decision;
	const unsigned long long* agree;
	size_t L;
	const unsigned long long* block_base;
	float* out;
};
__global__ void __launch_bounds__(1024) graded_emit(const GradedEmitArgs a) {
	__shared__ unsigned wave_n[16];
	const FuseView& R = a.views[a.ref];
	const size_t p = (size_t)blockIdx.x * 1024 + threadIdx.x;
	const bool acc = p < a.L && a.decision[p] == 1;
	const unsigned long long m = __ballot(acc);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_n[wave] = (unsigned)__popcll(m);
	__syncthreads();
	if (!acc) return;
	unsigned rank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
	for (int w = 0; w < wave; ++w) rank += wave_n[w];
	const int y = (int)(p / R.cols), x = (int)(p - (size_t)y * R.cols);
	const f3 X = fuse_lift(R, x, y, R.depth[p]);
	float sum[3] = { (float)R.bgr[3 * p], (float)R.bgr[3 * p + 1], (float)R.bgr[3 * p + 2] };
	if (!a.advanced) {
		const unsigned long long agree = a.agree[p];
		int count = 0;
		for (int j = 0; j < a.ns; ++j) {
			if (!((agree >> j) & 1)) continue;
			// the colour under the residual in force: the source pixel of the comparison the residual came from (APD.cpp:2101-2108)
			const FuseView& S = a.views[a.src[j]];
			const int q = a.last_idx[(size_t)j * a.L + p];
			const int sp = a.f_pix[(size_t)j * a.L + q];
			int sx = sp % S.cols, sy = sp / S.cols;
			sx = sx < S.cols - 1 ? sx : S.cols - 1;
			sy = sy < S.rows - 1 ? sy : S.rows - 1;
			const size_t si = (size_t)sy * S.cols + sx;
			sum[0] += S.bgr[3 * si]; sum[1] += S.bgr[3 * si + 1]; sum[2] += S.bgr[3 * si + 2];
			++count;
		}
		sum[0] /= (count + 1.0f); sum[1] /= (count + 1.0f); sum[2] /= (count + 1.0f);
	}
	float* o = a.out + (a.block_base[blockIdx.x] + rank) * 6;
	o[0] = X.x; o[1] = X.y; o[2] = X.z; o[3] = sum[0]; o[4] = sum[1]; o[5] = sum[2];
	R.claimed[p] = 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// the job
// ------------------------------------------------------------------------------------------------
struct dvp_fuse {
	int device = 0;
	int num_views = 0;
	hipStream_t stream = nullptr;
	std::vector<FuseView> views;          // host copies (device pointers inside)
	std::vector<char> have;
	FuseView* views_dev = nullptr;
	bool views_dirty = true;
	std::vector<void*> allocs;
	// per-call scratch (grown on demand)
	size_t cap_L = 0;
	int cap_ns = 0;
	int *cand_view = nullptr, *cand_pix = nullptr, *src_dev = nullptr;
	float* cand_vote = nullptr;
	signed char* count = nullptr;
	uint8_t* decision = nullptr;
	unsigned long long* live = nullptr;
	unsigned *list_a = nullptr, *list_b = nullptr, *counters = nullptr, *block_count = nullptr;
	unsigned long long* block_base = nullptr;
	// the graded variants' extra arrays
	float *g_rel = nullptr, *g_ang = nullptr;
	int *g_last = nullptr, *g_block_last = nullptr, *g_block_carry = nullptr;
	size_t g_cap_L = 0;
	int g_cap_ns = 0;
	unsigned round_serial = 0, view_serial = 0;
	// the cloud: one device block per fused view
	struct Segment { float* dev; long long n; };
	std::vector<Segment> segments;
	long long total = 0;
	int last_rounds = 0, last_rest = 0;
	std::string error;
};

static std::string g_fuse_create_error;
#define FUSE_TRY(f, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (f)->error = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; } } while (0)

extern "C" {

int dvp_fuse_create(int device, int num_views, dvp_fuse** out) {
	if (!out || num_views <= 0) { g_fuse_create_error = "dvp_fuse_create: bad arguments"; return 1; }
	if (hipSetDevice(device) != hipSuccess) { g_fuse_create_error = "dvp_fuse_create: hipSetDevice failed (no GPU?)"; return 1; }
	dvp_fuse* f = new dvp_fuse;
	f->device = device;
	f->num_views = num_views;
	f->views.resize(num_views);
	f->have.assign(num_views, 0);
	if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&f->views_dev, sizeof(FuseView) * num_views) != hipSuccess) {
		g_fuse_create_error = "dvp_fuse_create: stream / allocation failed";
		delete f;
		return 1;
	}
	*out = f;
	return 0;
}

int dvp_fuse_destroy(dvp_fuse* f) {
	if (!f) return 0;
	(void)hipSetDevice(f->device);
	if (f->stream) (void)hipStreamSynchronize(f->stream);
	for (void* p : f->allocs) (void)hipFree(p);
	for (auto& s : f->segments) (void)hipFree(s.dev);
	(void)hipFree(f->views_dev);
	for (void* p : { (void*)f->g_rel, (void*)f->g_ang, (void*)f->g_last, (void*)f->g_block_last, (void*)f->g_block_carry })
		if (p) (void)hipFree(p);
	for (void* p : { (void*)f->cand_view, (void*)f->cand_pix, (void*)f->cand_vote, (void*)f->count, (void*)f->decision, (void*)f->live, (void*)f->list_a, (void*)f->list_b,
	                 (void*)f->counters, (void*)f->block_count, (void*)f->block_base, (void*)f->src_dev })
		if (p) (void)hipFree(p);
	if (f->stream) (void)hipStreamDestroy(f->stream);
	delete f;
	return 0;
}

const char* dvp_fuse_last_error(const dvp_fuse* f) { return f ? f->error.c_str() : g_fuse_create_error.c_str(); }

// The maps of view slot `v` (host pointers; copied).  `cam` is the camera AS THE FUSION SEES IT: intrinsics already rescaled to
// the maps' size (RescaleImageAndCamera, APD.cpp:1750-1771).  weak == null: every pixel STRONG; block == null: no mask.
int dvp_fuse_set_view(dvp_fuse* f, int v, const DvpCamera* cam, int cols, int rows, const float* depth, const float* normal_xyz,
                      const uint8_t* weak, const uint8_t* bgr, const uint8_t* block) {
	if (!f) return 1;
	if (v < 0 || v >= f->num_views || !cam || !depth || !normal_xyz || !bgr || cols <= 0 || rows <= 0) { f->error = "dvp_fuse_set_view: bad arguments"; return 1; }
	if ((size_t)cols * rows >= ((size_t)1 << 31)) { f->error = "dvp_fuse_set_view: more than 2^31 pixels"; return 1; }
	if (f->have[v]) { f->error = "dvp_fuse_set_view: the view was set before"; return 1; }
	FUSE_TRY(f, hipSetDevice(f->device));
	const size_t L = (size_t)cols * rows;
	FuseView fv;
	std::memset(&fv, 0, sizeof(fv));
	fv.cam = *cam;
	for (int k = 0; k < 3; ++k) fv.centre[k] = -(cam->R[0 + k] * cam->t[0] + cam->R[3 + k] * cam->t[1] + cam->R[6 + k] * cam->t[2]);
	fv.cols = cols;
	fv.rows = rows;
	auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
		void* p = nullptr;
		if (hipMalloc(&p, bytes) != hipSuccess) { f->error = "dvp_fuse_set_view: out of device memory"; return 1; }
		f->allocs.push_back(p);
		if (hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, f->stream) != hipSuccess) { f->error = "dvp_fuse_set_view: upload failed"; return 1; }
		*dst = p;
		return 0;
	};
	if (up(depth, L * 4, (const void**)&fv.depth) || up(normal_xyz, L * 12, (const void**)&fv.normal) || up(bgr, L * 3, (const void**)&fv.bgr)) return 1;
	if (weak && up(weak, L, (const void**)&fv.weak)) return 1;
	if (block && up(block, L, (const void**)&fv.block)) return 1;
	void *cl = nullptr, *und = nullptr, *acc = nullptr;
	if (hipMalloc(&cl, L) != hipSuccess || hipMalloc(&und, L * 8) != hipSuccess || hipMalloc(&acc, L * 8) != hipSuccess) { f->error = "dvp_fuse_set_view: out of device memory"; return 1; }
	f->allocs.push_back(cl); f->allocs.push_back(und); f->allocs.push_back(acc);
	FUSE_TRY(f, hipMemsetAsync(cl, 0, L, f->stream));
	FUSE_TRY(f, hipMemsetAsync(und, 0xFF, L * 8, f->stream));
	FUSE_TRY(f, hipMemsetAsync(acc, 0xFF, L * 8, f->stream));
	fv.claimed = (uint8_t*)cl;
	fv.und = (unsigned long long*)und;
	fv.acc = (unsigned long long*)acc;
	FUSE_TRY(f, hipStreamSynchronize(f->stream));   // the caller's buffers are free again
	f->views[v] = fv;
	f->have[v] = 1;
	f->views_dirty = true;
	return 0;
}

static int fuse_reserve(dvp_fuse* f, size_t L, int ns) {
	if (L <= f->cap_L && ns <= f->cap_ns) return 0;
	const size_t nL = std::max(L, f->cap_L);
	const int nn = std::max(ns, f->cap_ns);
	for (void** p : { (void**)&f->cand_view, (void**)&f->cand_pix, (void**)&f->cand_vote, (void**)&f->count, (void**)&f->decision, (void**)&f->live, (void**)&f->list_a,
	                  (void**)&f->list_b, (void**)&f->counters, (void**)&f->block_count, (void**)&f->block_base, (void**)&f->src_dev }) {
		if (*p) (void)hipFree(*p);
		*p = nullptr;
	}
	f->cap_L = 0;
	f->cap_ns = 0;
	const size_t blocks = (nL + 1023) / 1024;
	if (hipMalloc((void**)&f->cand_view, (size_t)nn * nL * 4) != hipSuccess || hipMalloc((void**)&f->cand_pix, (size_t)nn * nL * 4) != hipSuccess ||
	    hipMalloc((void**)&f->cand_vote, (size_t)nn * nL * 4) != hipSuccess || hipMalloc((void**)&f->count, nL) != hipSuccess ||
	    hipMalloc((void**)&f->decision, nL) != hipSuccess || hipMalloc((void**)&f->live, nL * 8) != hipSuccess ||
	    hipMalloc((void**)&f->list_a, nL * 4) != hipSuccess || hipMalloc((void**)&f->list_b, nL * 4) != hipSuccess ||
	    hipMalloc((void**)&f->counters, 16) != hipSuccess || hipMalloc((void**)&f->block_count, blocks * 4) != hipSuccess ||
	    hipMalloc((void**)&f->block_base, blocks * 8) != hipSuccess || hipMalloc((void**)&f->src_dev, (size_t)std::max(nn, 1) * 4) != hipSuccess) {
		(void)hipGetLastError();
		f->error = "dvp_fuse_view: out of device memory for the candidate lists";
		return 1;
	}
	f->cap_L = nL;
	f->cap_ns = nn;
	return 0;
}

// A round is two small launches and one 4-byte read-back (~30 us); a long dependency chain — pixels along a row sharing witnesses
// pairwise — shrinks the undecided list slowly in its tail.  The rounds go on while the list is worth a launch; a short rest
// (or a chain longer than the round limit) is finished in index order by one lane (~2 us per pixel).
#ifndef DVP_FUSE_MAX_ROUNDS
#define DVP_FUSE_MAX_ROUNDS 4096
#endif
#ifndef DVP_FUSE_REST
#define DVP_FUSE_REST 96
#endif

// Fuse view slot `v` against the source slots `src` (pair.txt order; slots without maps must have been left out by the caller,
// as RunFusion skips them): the accepted points are appended to the cloud in scan order.
int dvp_fuse_view(dvp_fuse* f, int v, const int* src, int num_src) {
	if (!f) return 1;
	if (v < 0 || v >= f->num_views || !f->have[v] || num_src < 0 || (num_src > 0 && !src)) { f->error = "dvp_fuse_view: bad arguments"; return 1; }
	if (num_src > 64) { f->error = "dvp_fuse_view: more than 64 source views"; return 1; }
	for (int j = 0; j < num_src; ++j)
		if (src[j] < 0 || src[j] >= f->num_views || !f->have[src[j]] || src[j] == v) { f->error = "dvp_fuse_view: bad source slot"; return 1; }
	FUSE_TRY(f, hipSetDevice(f->device));
	if (f->views_dirty) {
		FUSE_TRY(f, hipMemcpyAsync(f->views_dev, f->views.data(), sizeof(FuseView) * f->num_views, hipMemcpyHostToDevice, f->stream));
		FUSE_TRY(f, hipStreamSynchronize(f->stream));
		f->views_dirty = false;
	}
	const FuseView& R = f->views[v];
	const size_t L = (size_t)R.cols * R.rows;
	if (fuse_reserve(f, L, std::max(num_src, 1))) return 1;
	if (num_src > 0) FUSE_TRY(f, hipMemcpyAsync(f->src_dev, src, (size_t)num_src * 4, hipMemcpyHostToDevice, f->stream));
	const size_t capL = f->cap_L;   // stride of the candidate arrays is the view's own L (passed as a.L below), the capacity only bounds it
	(void)capL;
	GatherArgs ga;
	ga.views = f->views_dev; ga.ref = v; ga.ns = num_src; ga.src = f->src_dev;
	ga.cand_view = f->cand_view; ga.cand_pix = f->cand_pix; ga.cand_vote = f->cand_vote; ga.count = f->count;
	const unsigned gL = (unsigned)((L + 255) / 256);
	hipLaunchKernelGGL(fuse_gather, dim3(gL), dim3(256), 0, f->stream, ga);
	FUSE_TRY(f, hipMemsetAsync(f->decision, 0, L, f->stream));
	FUSE_TRY(f, hipMemsetAsync(f->counters, 0, 16, f->stream));
	hipLaunchKernelGGL(fuse_first_list, dim3(gL), dim3(256), 0, f->stream, f->count, L, f->list_a, f->counters);
	FUSE_TRY(f, hipGetLastError());
	unsigned n_list = 0;
	FUSE_TRY(f, hipMemcpyAsync(&n_list, f->counters, 4, hipMemcpyDeviceToHost, f->stream));
	FUSE_TRY(f, hipStreamSynchronize(f->stream));
	ResolveArgs ra;
	ra.views = f->views_dev; ra.ref = v; ra.ns = num_src;
	ra.cand_view = f->cand_view; ra.cand_pix = f->cand_pix; ra.cand_vote = f->cand_vote; ra.count = f->count;
	ra.decision = f->decision; ra.live = f->live; ra.L = L;
	ra.view_hi = ~(++f->view_serial);
	unsigned* cur = f->list_a;
	unsigned* nxt = f->list_b;
	int rounds = 0;
	f->last_rest = 0;
	while (n_list > 0) {
		if (rounds >= DVP_FUSE_MAX_ROUNDS || (rounds > 0 && n_list <= DVP_FUSE_REST)) {
			// a long dependency chain (many pixels in a row sharing witnesses pairwise): the rest in index order by one lane
			std::vector<unsigned> rest(n_list);
			FUSE_TRY(f, hipMemcpy(rest.data(), cur, (size_t)n_list * 4, hipMemcpyDeviceToHost));
			std::sort(rest.begin(), rest.end());
			FUSE_TRY(f, hipMemcpy(cur, rest.data(), (size_t)n_list * 4, hipMemcpyHostToDevice));
			ra.list = cur; ra.n_list = n_list; ra.next = nullptr; ra.n_next = nullptr; ra.round_hi = 0;
			hipLaunchKernelGGL(fuse_decide_rest, dim3(1), dim3(64), 0, f->stream, ra);
			FUSE_TRY(f, hipGetLastError());
			f->last_rest = (int)n_list;
			break;
		}
		ra.round_hi = ~(++f->round_serial);
		ra.list = cur; ra.n_list = n_list; ra.next = nxt; ra.n_next = f->counters + 1;
		FUSE_TRY(f, hipMemsetAsync(f->counters + 1, 0, 4, f->stream));
		const unsigned g = (n_list + 255) / 256;
		hipLaunchKernelGGL(fuse_mark, dim3(g), dim3(256), 0, f->stream, ra);
		hipLaunchKernelGGL(fuse_decide, dim3(g), dim3(256), 0, f->stream, ra);
		FUSE_TRY(f, hipGetLastError());
		unsigned n_next = 0;
		FUSE_TRY(f, hipMemcpyAsync(&n_next, f->counters + 1, 4, hipMemcpyDeviceToHost, f->stream));
		FUSE_TRY(f, hipStreamSynchronize(f->stream));
		if (n_next >= n_list) { f->error = "dvp_fuse_view: a resolve round decided nothing"; return 1; }   // (cannot happen: the smallest undecided index always decides)
		std::swap(cur, nxt);
		n_list = n_next;
		++rounds;
	}
	f->last_rounds = rounds;
	// accepted pixels in raster order
	EmitArgs ea;
	ea.views = f->views_dev; ea.ref = v; ea.cand_view = f->cand_view; ea.cand_pix = f->cand_pix; ea.count = f->count;
	ea.decision = f->decision; ea.live = f->live; ea.L = L; ea.block_count = f->block_count; ea.block_base = f->block_base; ea.out = nullptr;
	const unsigned blocks = (unsigned)((L + 1023) / 1024);
	hipLaunchKernelGGL(fuse_count, dim3(blocks), dim3(1024), 0, f->stream, ea);
	FUSE_TRY(f, hipGetLastError());
	std::vector<unsigned> bc(blocks);
	FUSE_TRY(f, hipMemcpyAsync(bc.data(), f->block_count, (size_t)blocks * 4, hipMemcpyDeviceToHost, f->stream));
	FUSE_TRY(f, hipStreamSynchronize(f->stream));
	std::vector<unsigned long long> base(blocks);
	unsigned long long n_points = 0;
	for (unsigned b = 0; b < blocks; ++b) { base[b] = n_points; n_points += bc[b]; }
	if (n_points > 0) {
		float* seg = nullptr;
		if (hipMalloc((void**)&seg, (size_t)n_points * 24) != hipSuccess) { (void)hipGetLastError(); f->error = "dvp_fuse_view: out of device memory for the points"; return 1; }
		f->segments.push_back(dvp_fuse::Segment{ seg, (long long)n_points });
		FUSE_TRY(f, hipMemcpyAsync(f->block_base, base.data(), (size_t)blocks * 8, hipMemcpyHostToDevice, f->stream));
		ea.out = seg;
		hipLaunchKernelGGL(fuse_emit, dim3(blocks), dim3(1024), 0, f->stream, ea);
		FUSE_TRY(f, hipGetLastError());
		FUSE_TRY(f, hipStreamSynchronize(f->stream));   // `base` is read by the copy above
		f->total += (long long)n_points;
	}
	return 0;
}

// One iteration of the outer loop of RunFusion_TAT_Intermediate (advanced = 0, APD.cpp:1962-2130) / RunFusion_TAT_advanced
// (advanced = 1, APD.cpp:2132-2279): `src` holds ALL sources of the view in pair.txt order, -1 for one without maps (it takes
// part in the count of sources the acceptance loop runs to, but never yields a comparison).
int dvp_fuse_view_graded(dvp_fuse* f, int v, const int* src, int num_src, int advanced) {
	if (!f) return 1;
	if (v < 0 || v >= f->num_views || !f->have[v] || num_src < 0 || (num_src > 0 && !src)) { f->error = "dvp_fuse_view_graded: bad arguments"; return 1; }
	if (num_src > 64) { f->error = "dvp_fuse_view_graded: more than 64 source views"; return 1; }
	for (int j = 0; j < num_src; ++j)
		if (src[j] >= f->num_views || (src[j] >= 0 && (!f->have[src[j]] || src[j] == v))) { f->error = "dvp_fuse_view_graded: bad source slot"; return 1; }
	FUSE_TRY(f, hipSetDevice(f->device));
	if (f->views_dirty) {
		FUSE_TRY(f, hipMemcpyAsync(f->views_dev, f->views.data(), sizeof(FuseView) * f->num_views, hipMemcpyHostToDevice, f->stream));
		FUSE_TRY(f, hipStreamSynchronize(f->stream));
		f->views_dirty = false;
	}
	const FuseView& R = f->views[v];
	const size_t L = (size_t)R.cols * R.rows;
	const int ns = std::max(num_src, 1);
	if (fuse_reserve(f, L, ns)) return 1;
	const unsigned gblocks = (unsigned)((L + kGradedBlock - 1) / kGradedBlock);
	if (L > f->g_cap_L || ns > f->g_cap_ns) {
		for (void** p : { (void**)&f->g_rel, (void**)&f->g_ang, (void**)&f->g_last, (void**)&f->g_block_last, (void**)&f->g_block_carry }) { if (*p) (void)hipFree(*p); *p = nullptr; }
		const size_t nL = std::max(L, f->g_cap_L);
		const int nn = std::max(ns, f->g_cap_ns);
		const size_t nb = (nL + kGradedBlock - 1) / kGradedBlock;
		f->g_cap_L = 0; f->g_cap_ns = 0;
		if (hipMalloc((void**)&f->g_rel, (size_t)nn * nL * 4) != hipSuccess || hipMalloc((void**)&f->g_ang, (size_t)nn * nL * 4) != hipSuccess || hipMalloc((void**)&f->g_last, (size_t)nn * nL * 4) != hipSuccess ||
		    hipMalloc((void**)&f->g_block_last, nb * nn * 4) != hipSuccess || hipMalloc((void**)&f->g_block_carry, nb * nn * 4) != hipSuccess) {
			(void)hipGetLastError();
			f->error = "dvp_fuse_view_graded: out of device memory";
			return 1;
		}
		f->g_cap_L = nL; f->g_cap_ns = nn;
	}
	if (num_src > 0) FUSE_TRY(f, hipMemcpyAsync(f->src_dev, src, (size_t)num_src * 4, hipMemcpyHostToDevice, f->stream));
	GradedArgs ga;
	ga.views = f->views_dev; ga.ref = v; ga.ns = num_src; ga.advanced = advanced ? 1 : 0; ga.src = f->src_dev;
	ga.f_err = f->cand_vote; ga.f_rel = f->g_rel; ga.f_ang = f->g_ang; ga.f_pix = f->cand_pix;
	ga.is_ref = reinterpret_cast<uint8_t*>(f->count); ga.block_last = f->g_block_last; ga.block_carry = f->g_block_carry;
	ga.last_idx = f->g_last; ga.decision = f->decision; ga.agree = f->live; ga.L = L;
	hipLaunchKernelGGL(graded_gather, dim3(gblocks), dim3(kGradedBlock), 0, f->stream, ga);
	FUSE_TRY(f, hipGetLastError());
	std::vector<int> last((size_t)gblocks * ns), carry((size_t)gblocks * ns);
	if (num_src > 0) {
		FUSE_TRY(f, hipMemcpyAsync(last.data(), f->g_block_last, (size_t)gblocks * num_src * 4, hipMemcpyDeviceToHost, f->stream));
		FUSE_TRY(f, hipStreamSynchronize(f->stream));
		std::vector<int> run(num_src, -1);
		for (unsigned b = 0; b < gblocks; ++b)
			for (int j = 0; j < num_src; ++j) {
				carry[(size_t)b * num_src + j] = run[j];
				run[j] = std::max(run[j], last[(size_t)b * num_src + j]);
			}
		FUSE_TRY(f, hipMemcpyAsync(f->g_block_carry, carry.data(), (size_t)gblocks * num_src * 4, hipMemcpyHostToDevice, f->stream));
	}
	hipLaunchKernelGGL(graded_decide, dim3(gblocks), dim3(kGradedBlock), 0, f->stream, ga);
	FUSE_TRY(f, hipGetLastError());
	EmitArgs ea;
	ea.views = f->views_dev; ea.ref = v; ea.cand_view = nullptr; ea.cand_pix = nullptr; ea.count = nullptr;
	ea.decision = f->decision; ea.live = nullptr; ea.L = L; ea.block_count = f->block_count; ea.block_base = f->block_base; ea.out = nullptr;
	const unsigned blocks = (unsigned)((L + 1023) / 1024);
	hipLaunchKernelGGL(fuse_count, dim3(blocks), dim3(1024), 0, f->stream, ea);
	FUSE_TRY(f, hipGetLastError());
	std::vector<unsigned> bc(blocks);
	FUSE_TRY(f, hipMemcpyAsync(bc.data(), f->block_count, (size_t)blocks * 4, hipMemcpyDeviceToHost, f->stream));
	FUSE_TRY(f, hipStreamSynchronize(f->stream));   // (also: `carry` has been read)
	std::vector<unsigned long long> base(blocks);
	unsigned long long n_points = 0;
	for (unsigned b = 0; b < blocks; ++b) { base[b] = n_points; n_points += bc[b]; }
	f->last_rounds = 0; f->last_rest = 0;
	if (n_points > 0) {
		float* seg = nullptr;
		if (hipMalloc((void**)&seg, (size_t)n_points * 24) != hipSuccess) { (void)hipGetLastError(); f->error = "dvp_fuse_view_graded: out of device memory for the points"; return 1; }
		f->segments.push_back(dvp_fuse::Segment{ seg, (long long)n_points });
		FUSE_TRY(f, hipMemcpyAsync(f->block_base, base.data(), (size_t)blocks * 8, hipMemcpyHostToDevice, f->stream));
		GradedEmitArgs ge;
		ge.views = f->views_dev; ge.ref = v; ge.ns = num_src; ge.advanced = advanced ? 1 : 0; ge.src = f->src_dev; ge.last_idx = f->g_last; ge.f_pix = f->cand_pix;
		ge.decision = f->decision; ge.agree = f->live; ge.L = L; ge.block_base = f->block_base; ge.out = seg;
		hipLaunchKernelGGL(graded_emit, dim3(blocks), dim3(1024), 0, f->stream, ge);
		FUSE_TRY(f, hipGetLastError());
		FUSE_TRY(f, hipStreamSynchronize(f->stream));
		f->total += (long long)n_points;
	}
	return 0;
}

long long dvp_fuse_count(const dvp_fuse* f) { return f ? f->total : -1; }

// resolve statistics of the last dvp_fuse_view: rounds of the parallel resolve, pixels left to the sequential finish
int dvp_fuse_last_rounds(const dvp_fuse* f, int* rounds, int* rest) {
	if (!f) return 1;
	if (rounds) *rounds = f->last_rounds;
	if (rest) *rest = f->last_rest;
	return 0;
}

// The cloud so far: dvp_fuse_count() records of six floats (x y z b g r: struct PointList, main.h:69-72) in scan order.
int dvp_fuse_download(dvp_fuse* f, float* points) {
	if (!f) return 1;
	if (!points && f->total > 0) { f->error = "dvp_fuse_download: null destination"; return 1; }
	FUSE_TRY(f, hipSetDevice(f->device));
	size_t off = 0;
	for (const auto& s : f->segments) {
		FUSE_TRY(f, hipMemcpyAsync(points + off, s.dev, (size_t)s.n * 24, hipMemcpyDeviceToHost, f->stream));
		off += (size_t)s.n * 6;
	}
	FUSE_TRY(f, hipStreamSynchronize(f->stream));
	return 0;
}

}  // extern "C"
