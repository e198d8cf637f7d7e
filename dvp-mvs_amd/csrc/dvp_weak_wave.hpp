// dvp_weak_wave.hpp — the weak-pixel update (CheckerboardPropagationWeak +
// PlaneHypothesisRefinementWeak, APD.cu:2739-3089, 1897-2008) with ONE WAVEFRONT PER WEAK PIXEL.
//
// Why not one lane per pixel (round 1): the deformable NCC of a WEAK pixel (ComputeBilateralNCCNew,
// APD.cu:835-1021) gathers, per (plane, view), 36 centre taps + up to 11 anchors x 9 taps around
// STRONG points that every pixel draws at random.  With a pixel per lane the 64 gathers of one load
// instruction land on 64 unrelated cache lines, nothing is reused before it is evicted, and PMC showed
// the kernel at the HBM roofline moving 64 B per 16-B footprint (6.7 G L2 misses per launch at
// 6208x4128, L2 hit rate 20 %).  Here the 64 lanes of a wave own the (plane, tap) pairs of one pixel:
// the <= 8 planes of a phase project a tap to neighbouring texels, so one load instruction touches a
// few lines instead of 64 and the lines are shared inside the instruction; the per-pixel state
// (cost vectors, weights, tables) lives in LDS instead of 4 KB of scratch per lane.
//
// Code shape.  A wave runs the function below in lock step.  `DVP_LANES(l) { ... }` marks a section in
// which lane l does its own share (on the device: the calling lane; in the host emulation of
// tests/emul: a loop over 64 lanes); everything outside such sections is per-pixel ("uniform") code
// that every lane executes with identical values, writing shared state only from lane 0
// (`DVP_LANE0`).  wave_sync() orders shared-memory traffic between sections.  All floating-point
// expressions, their order and their operands are those of the per-pixel formulation (the oracle's),
// only the assignment of independent work items to lanes is new.
#ifndef DVP_WEAK_WAVE_HPP_
#define DVP_WEAK_WAVE_HPP_

#include "dvp_weak.hpp"

#if defined(DVP_GN_STATS) && !defined(__HIPCC__)
#include <cstdio>
#endif
namespace dvp {

#if defined(__HIP_DEVICE_COMPILE__)
#define DVP_LANES(L) for (int L = (int)(threadIdx.x & 63u), once_##L = 1; once_##L; once_##L = 0)
#define DVP_LANE0 ((threadIdx.x & 63u) == 0u)
DVP_HD void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#else
#define DVP_LANES(L) for (int L = 0; L < 64; ++L)
#define DVP_LANE0 true
DVP_HD void wave_sync() {}
#endif

constexpr int kAnchors = DVP_NEIGHBOUR_NUM - 1;   // 11
constexpr int kAnchorTaps = kAnchors * 9;         // 99
#ifndef DVP_WEAK_PAIRS
#define DVP_WEAK_PAIRS 40
#endif
// (source view, plane) pairs per shared-memory hand-over of the weak update: a batch takes as many views as fit,
// 5 with the 8 candidate planes, 20 / 8 with the 2 and 5 planes of the refinement phases (8.6 KB of LDS per wave:
// four workgroups per CU)
constexpr int kWeakPairs = DVP_WEAK_PAIRS;
constexpr int kWeakViews = 7;   // ... and at most this many views (size of the (view, anchor) prefetch table)

// per-wave shared state (LDS on the device).  TAB = 1 (the anchor reference sides come from the pass' table): no offset
// table, and the homographies of the batch's (view, plane) pairs instead — in the bytes of cost_array, which is dead
// while a batch is evaluated (written by the epilogue of the propagation phase, read by the view selection).
template <int TAB>
struct WeakSharedT {
	f2 ctab[kTaps * kTaps];        // centre patch: (w, w*ref) per tap, row-major
	float caa[kTaps * kTaps];      // w*ref*ref per tap (reference moments only)
	float rows[kWeakPairs][kTaps][3];    // centre-patch row sums (s_s, s_ss, s_rs) per pair (view slot * np + plane) and row; the final-cost section uses pairs 0..7 for 8 views
	float acost[kWeakPairs][kAnchors];   // anchor cost per (pair, anchor), < 0: does not count
	int inq[kWeakPairs];                 // pair: the centre projects inside the source image
	// (view slot, anchor) of the batch, fetched once before the items run: the anchor's 8 visibility-prior offsets
	// for that view as signed bytes (x, y) and what anchor_cost has to do with the pair
	uint32_t aoff[TAB ? 1 : kWeakViews * kAnchors][4];
	uint8_t astate[kWeakViews * kAnchors];   // 0: no anchor; 1: the anchor did not select the view; 2: sub-patch
	union {
		float cost_array[8][32];
		float Hq[TAB ? kWeakPairs : 1][9];   // homography of the batch's pair (view slot * np + live-plane index)
	};
	float ev[8][32];
	float gtab[8][32];             // geometric-consistency cost per (plane, view)
	float probs[32];
	float priors[32];
	float fcost[8];
	int positions[8];
	uint8_t vw[32];
	f4 pl[8];
};
using WeakShared = WeakSharedT<0>;

// build_patch_ctx (dvp_ncc.hpp) by the wave: lane t < 36 owns tap t; the reference moments are then
// summed in the row-then-total order by every lane.
template <int FMT, class SH>
DVP_HD void wave_patch_ctx(const Dev& d, int px, int py, int radius, int inc, int colour_only, SH& sh, PatchCtx* c) {
	c->radius = radius;
	c->inc = inc;
	c->fast = (inc > 0 && (2 * radius) / inc + 1 == kTaps) ? 1 : 0;
	c->sum_ref = c->sum_ref_ref = c->wsum = 0.0f;
	if (!c->fast) return;
	const int W = d.width, H = d.height, P = d.pitch;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	DVP_LANES(t) {
		if (t >= kTaps * kTaps) continue;
		const int ty = t / kTaps, tx = t - ty * kTaps;
		const int i = -radius + tx * inc, j = -radius + ty * inc;
		const float a = ref_texel_t<FMT>(d, px + i, py + j);
		const float w = bilateral_weight((float)i, (float)j, a, cpix, d.params.sigma_spatial, d.params.sigma_color, colour_only);
		const float wa = w * a;
		sh.ctab[t] = mk2(w, wa);
		sh.caa[t] = wa * a;
	}
	wave_sync();
	float sr = 0.0f, srr = 0.0f, ws = 0.0f;
	for (int ty = 0; ty < kTaps; ++ty) {
		float sr_row = 0.0f, srr_row = 0.0f, ws_row = 0.0f;
		for (int tx = 0; tx < kTaps; ++tx) {
			const f2 t = sh.ctab[ty * kTaps + tx];
			sr_row += t.y;
			srr_row += sh.caa[ty * kTaps + tx];
			ws_row += t.x;
		}
		sr += sr_row;
		srr += srr_row;
		ws += ws_row;
	}
	c->sum_ref = sr;
	c->sum_ref_ref = srr;
	c->wsum = ws;
}

// one row (6 taps) of the 36-tap patch for homography H: the row's three source-side sums
// (ncc_patch_fast, dvp_ncc.hpp: same products, same shared division per row, same order)
template <int SMP, int FMT, class SH>
DVP_HD void patch_row_sums(const Dev& d, const SH& sh, const float* H, const void* src, int px, int py, int radius, int inc, int row, float* out /*[3]*/) {
	const int W = d.width, Hh = d.height, P = d.pitch;
	const float fy = (float)(py - radius + row * inc);
	const float hy1 = H[1] * fy, hy4 = H[4] * fy, hy7 = H[7] * fy;
	float X[kTaps], Y[kTaps], Z[kTaps], IZ[kTaps];
#pragma unroll
	for (int tx = 0; tx < kTaps; ++tx) {
		const float fx = (float)(px - radius + tx * inc);
		X[tx] = H[0] * fx + hy1 + H[2];
		Y[tx] = H[3] * fx + hy4 + H[5];
		Z[tx] = H[6] * fx + hy7 + H[8];
	}
	batch_rcp(Z, kTaps, IZ);
	unsigned off[kTaps];
	TapW<SMP> tw[kTaps];
	float q[kTaps][4];
#pragma unroll
	for (int tx = 0; tx < kTaps; ++tx) tex_coord_t<FMT>(d, X[tx] * IZ[tx], Y[tx] * IZ[tx], &off[tx], &tw[tx]);
#pragma unroll
	for (int tx = 0; tx < kTaps; ++tx) load_quad_t<FMT>(src, off[tx], &q[tx][0], &q[tx][1], &q[tx][2], &q[tx][3]);
	float r_s = 0.0f, r_ss = 0.0f, r_rs = 0.0f;
#pragma unroll
	for (int tx = 0; tx < kTaps; ++tx) {
		float fa, fb;
		tap_weights(tw[tx], &fa, &fb);
		const float b = tex_lerp(fa, fb, q[tx][0], q[tx][1], q[tx][2], q[tx][3]);
		const f2 t = sh.ctab[row * kTaps + tx];
		const float wsb = t.x * b;
		r_s += wsb;
		r_ss = fmaf(wsb, b, r_ss);
		r_rs = fmaf(t.y, b, r_rs);
	}
	out[0] = r_s;
	out[1] = r_ss;
	out[2] = r_rs;
}

// index of the n-th set bit of m (n < popcount(m))
DVP_HD int nth_set_bit(uint32_t m, int n) {
	for (int i = 0; i < n; ++i) m &= m - 1;
	return __builtin_ctz(m);
}

// One anchor sub-patch of ComputeBilateralNCCNew (APD.cu:905-1000): the anchor `nb` of the pixel against source
// view v (1-based) under homography H.  < 0: the anchor does not count.  The anchor's 9 reference taps (the 8
// visibility-prior offsets of (anchor, view) + the anchor itself: offsets, texels, weights, sums) and the 9 gathers
// in the source image all stay in registers.
template <int SMP, int FMT>
DVP_HD float anchor_cost(const Dev& d, const float* H, const void* src, s2 nb, int state, const uint32_t* offs, float cpix) {
	const int W = d.width, Hh = d.height, Pt = d.pitch;
	if (state == 0) return -1.0f;
	const bool visible = state == 2;
	const f2 nsp = apply_homography(H, nb.x, nb.y);
	const bool outside = nsp.x < 0 || nsp.y < 0 || nsp.x >= W || nsp.y >= Hh;
	if (outside) return visible ? 2.0f : -1.0f;
	if (!visible) return 2.0f;   // anchor not visible in this view: the reference's 0/0 path yields exactly 2
	int tx[9], ty[9];
	float ti[9], tj[9];
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		int i = 0, j = 0;
		if (t < 8) {
			const uint32_t o = offs[t >> 1] >> (16 * (t & 1));
			i = (int)(int8_t)(o & 255u);
			j = (int)(int8_t)((o >> 8) & 255u);
			if (i == 0 && j == 0) {   // default +-5 ring (APD.cu:943-952): {-5,0,5}^2 without its centre, x-major
				const int u = t + (t >= 4 ? 1 : 0);
				i = (u / 3 - 1) * 5;
				j = (u % 3 - 1) * 5;
			}
		}
		tx[t] = nb.x + i;
		ty[t] = nb.y + j;
		ti[t] = (float)i;
		tj[t] = (float)j;
	}
	// source side first (addresses need only the offsets): 9 gathers in flight
	unsigned off[9];
	TapW<SMP> tw[9];
	float qd[9][4];
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		const f2 sp = apply_homography(H, tx[t], ty[t]);
		tex_coord_t<FMT>(d, sp.x, sp.y, &off[t], &tw[t]);
	}
#pragma unroll
	for (int t = 0; t < 9; ++t) load_quad_t<FMT>(src, off[t], &qd[t][0], &qd[t][1], &qd[t][2], &qd[t][3]);
	float av[9];
#pragma unroll
	for (int t = 0; t < 9; ++t) av[t] = ref_texel_t<FMT>(d, tx[t], ty[t]);
	float a_sr = 0.0f, a_srr = 0.0f, a_sw = 0.0f;
	float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		const float w = bilateral_weight(ti[t], tj[t], av[t], cpix, d.params.sigma_spatial, d.params.sigma_color, 1);
		const float wa = w * av[t];
		a_sr += wa;
		a_srr += wa * av[t];
		a_sw += w;
		float fa, fb;
		tap_weights(tw[t], &fa, &fb);
		const float b = tex_lerp(fa, fb, qd[t][0], qd[t][1], qd[t][2], qd[t][3]);
		const float wb = w * b;
		s_s += wb;
		s_ss = fmaf(wb, b, s_ss);
		s_rs = fmaf(wa, b, s_rs);
	}
	return ncc_from_sums(a_sr, a_srr, s_s, s_ss, s_rs, a_sw);
}

// ---- the anchor sub-patches' reference side, once per pass -------------------------------------------------------------
// Of an anchor sub-patch (APD.cu:936-1006) only the nine gathers in the source image and the three source-side sums
// depend on the plane hypothesis.  The nine tap positions (anchor + the visibility-prior offsets of (anchor, view)), the
// colour weights w = exp(-|I(tap) - I(pixel)| / 2 sigma_c^2), the products w * I(tap) and the three reference sums depend
// on (WEAK pixel, view, anchor) only — and anchors (GenNeighbours / NeigbourUpdate) and offsets (GenEdgeInform) are fixed
// before the first iteration.  anchor_cost() recomputes them for every (view, anchor, PLANE) item of all three phases of
// every weak-update launch: 9 reference texel loads, 9 exp and 27 multiply-adds per item.  build_anchor_record() forms
// them ONCE per pass — same operations, same order, hence the same bits — into one 128-byte record per (WEAK pixel, view,
// anchor) (Dev::anchor_tab: 12.4 KB per WEAK pixel at S = 9; 22 GB at 6208x4128 with 7 % WEAK — it is a 288 GB part), and
// anchor_cost_tab() reads the record with eight 16-byte loads that the eight plane lanes of a pair share.
struct alignas(16) AnchorRec {
	f2 wt[9];            // (w, w * ref) per tap
	uint32_t pos[9];     // tap position: (x & 0xffff) | (y << 16), signed 16-bit halves
	float a_sr, a_srr, a_sw;
	uint32_t pad[2];
};
static_assert(sizeof(AnchorRec) == 128, "one cache line per (WEAK pixel, view, anchor)");
DVP_HD size_t anchor_rec_index(const Dev& d, int weak_index, int v0, int k) {
	return ((size_t)weak_index * (size_t)(d.params.num_images - 1) + (size_t)v0) * kAnchors + (size_t)k;
}
// record of anchor k (0-based: neighbours[k + 1]) of the WEAK pixel `center` for source view v0 (0-based); FMT: the plane format
// the reference texels are read from (the same values either way)
// -> *out, *index (its place in Dev::anchor_tab); false: no anchor, the record is never read (state 0)
template <int FMT>
DVP_HD bool make_anchor_record(const Dev& d, int center, int v0, int k, AnchorRec* out, size_t* index) {
	const int W = d.width;
	const int py = center / W, px = center - py * W;
	const int wi = d.neighbours_map[center];
	const s2 nb = d.neighbours[(size_t)wi * DVP_NEIGHBOUR_NUM + k + 1];
	if (nb.x == -1 || nb.y == -1) return false;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	// the anchor's eight visibility-prior offsets for the view: one 32-byte record, two 16-byte loads (as eight 4-byte loads they
	// were 8 of the thread's 20 L2 requests, and the launch issues 160 G requests/s: PMC r05)
	s2 cand[8];
	{
		const s2* cp = d.candidate + cand_index(d, nb.x + nb.y * W, v0);
#if defined(__HIP_DEVICE_COMPILE__)
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		const u4 lo = reinterpret_cast<const u4*>(cp)[0], hi = reinterpret_cast<const u4*>(cp)[1];
		const uint32_t w[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
#pragma unroll
		for (int t = 0; t < 8; ++t) cand[t] = mks2((int)(int16_t)(w[t] & 0xffffu), (int)(int16_t)(w[t] >> 16));
#else
		for (int t = 0; t < 8; ++t) cand[t] = cp[t];
#endif
	}
	AnchorRec& r = *out;
	float a_sr = 0.0f, a_srr = 0.0f, a_sw = 0.0f;
	// the nine texels first (one round trip; fetched inside the tap loop they were nine, each behind the `default ring` branch)
	int ti[9], tj[9];
	float tav[9];
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		int i = 0, j = 0;
		if (t < 8) {
			i = cand[t].x;
			j = cand[t].y;
			const bool ring = i == 0 && j == 0;   // default +-5 ring (APD.cu:943-952): {-5,0,5}^2 without its centre, x-major
			const int u = t + (t >= 4 ? 1 : 0);
			i = ring ? (u / 3 - 1) * 5 : i;
			j = ring ? (u % 3 - 1) * 5 : j;
		}
		ti[t] = i; tj[t] = j;
		tav[t] = ref_texel_t<FMT>(d, nb.x + i, nb.y + j);
	}
	sched_fence();
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		const int i = ti[t], j = tj[t];
		const int tx = nb.x + i, ty = nb.y + j;
		const float av = tav[t];
		const float w = bilateral_weight((float)i, (float)j, av, cpix, d.params.sigma_spatial, d.params.sigma_color, 1);
		const float wa = w * av;
		a_sr += wa;
		a_srr += wa * av;
		a_sw += w;
		r.wt[t] = mk2(w, wa);
		r.pos[t] = ((uint32_t)tx & 0xffffu) | ((uint32_t)ty << 16);
	}
	r.a_sr = a_sr;
	r.a_srr = a_srr;
	r.a_sw = a_sw;
	r.pad[0] = r.pad[1] = 0u;
	*index = anchor_rec_index(d, wi, v0, k);
	return true;
}
template <int FMT = 0>
DVP_HD void build_anchor_record(const Dev& d, int center, int v0, int k) {
	AnchorRec r;
	size_t index;
	if (!make_anchor_record<FMT>(d, center, v0, k, &r, &index)) return;
#if defined(__HIP_DEVICE_COMPILE__)
	{   // eight 16-byte stores
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		const u4* src4 = reinterpret_cast<const u4*>(&r);
		u4* dst4 = reinterpret_cast<u4*>(d.anchor_tab + index);
#pragma unroll
		for (int i = 0; i < 8; ++i) dst4[i] = src4[i];
	}
#else
	d.anchor_tab[index] = r;
#endif
}

// anchor_cost() with the reference side taken from the pass' table
template <int SMP, int FMT>
DVP_HD float anchor_cost_tab(const Dev& d, const float* H, const void* src, s2 nb, int state, const AnchorRec* recp) {
	const int W = d.width, Hh = d.height, Pt = d.pitch;
	if (state == 0) return -1.0f;
	const bool visible = state == 2;
	const f2 nsp = apply_homography(H, nb.x, nb.y);
	const bool outside = nsp.x < 0 || nsp.y < 0 || nsp.x >= W || nsp.y >= Hh;
	if (outside) return visible ? 2.0f : -1.0f;
	if (!visible) return 2.0f;   // anchor not visible in this view: the reference's 0/0 path yields exactly 2
#if defined(DVP_ABL_NO_ANCHOR)   // timing ablation (wrong results): what the launch costs without the anchor sub-patches
	return 1.0f;
#endif
	AnchorRec r;
#if defined(__HIP_DEVICE_COMPILE__)
	{   // eight 16-byte loads (a plain struct copy is split into 30 dword loads)
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		const u4* src4 = reinterpret_cast<const u4*>(recp);
		u4* dst4 = reinterpret_cast<u4*>(&r);
#pragma unroll
		for (int i = 0; i < 8; ++i) dst4[i] = src4[i];
	}
#else
	r = *recp;
#endif
	unsigned off[9];
	TapW<SMP> tw[9];
	float qd[9][4];
#if defined(DVP_ABL_ANCHOR_BATCH_RCP)
	// TIMING A/B OF A CONTRACT CHANGE (different bits): the nine projective divides of a sub-patch through ONE
	// correctly rounded division (prefix products, as batch_rcp does for the six taps of a centre-patch row) instead of one each
	{
		float X[9], Y[9], Z[9], P[9], IZ[9];
#pragma unroll
		for (int t = 0; t < 9; ++t) {
			const int tx = (int)(int16_t)(r.pos[t] & 0xffffu), ty = (int)(int16_t)(r.pos[t] >> 16);
			X[t] = H[0] * tx + H[1] * ty + H[2];
			Y[t] = H[3] * tx + H[4] * ty + H[5];
			Z[t] = H[6] * tx + H[7] * ty + H[8];
		}
		P[0] = Z[0];
#pragma unroll
		for (int t = 1; t < 9; ++t) P[t] = P[t - 1] * Z[t];
		float rr = 1.0f / P[8];
#pragma unroll
		for (int t = 8; t >= 1; --t) { IZ[t] = rr * P[t - 1]; rr = rr * Z[t]; }
		IZ[0] = rr;
#pragma unroll
		for (int t = 0; t < 9; ++t) tex_coord_t<FMT>(d, X[t] * IZ[t], Y[t] * IZ[t], &off[t], &tw[t]);
	}
#else
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		const f2 sp = apply_homography(H, (int)(int16_t)(r.pos[t] & 0xffffu), (int)(int16_t)(r.pos[t] >> 16));
		tex_coord_t<FMT>(d, sp.x, sp.y, &off[t], &tw[t]);
#if defined(DVP_ABL_ANCHOR_NOGATHER)   // timing ablation (wrong results): every gather hits one line
		off[t] = (unsigned)t * 4u;
#endif
	}
#endif
#pragma unroll
	for (int t = 0; t < 9; ++t) load_quad_t<FMT>(src, off[t], &qd[t][0], &qd[t][1], &qd[t][2], &qd[t][3]);
	float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
#pragma unroll
	for (int t = 0; t < 9; ++t) {
		float fa, fb;
		tap_weights(tw[t], &fa, &fb);
		const float b = tex_lerp(fa, fb, qd[t][0], qd[t][1], qd[t][2], qd[t][3]);
		const float wb = r.wt[t].x * b;
		s_s += wb;
		s_ss = fmaf(wb, b, s_ss);
		s_rs = fmaf(r.wt[t].y, b, s_rs);
	}
	return ncc_from_sums(r.a_sr, r.a_srr, s_s, s_ss, s_rs, r.a_sw);
}

// ComputeBilateralNCCNew (APD.cu:835-1021) for the live planes sh.pl[q] (q in pmask) and the source views
// in vmask: sh.ev[q][view] = cost.  c = centre-patch context (colour-only weights) built by wave_patch_ctx.
// Views are taken kWeakPairs / np at a time with ONE shared-memory hand-over per batch:
//   section 1  per view of the batch, back to back:
//              lane (plane q, anchor k): the whole anchor sub-patch of that pair in registers — the
//              anchor's 9 reference taps (offsets, texels, weights, sums: identical in the 8 plane lanes
//              of an anchor, which costs nothing in lock step) and the 9 gathers in the source image;
//              the 8 planes of one tap sit in adjacent lanes of ONE load instruction and share lines.
//              The anchor pixel and its view mask do not depend on the view and are fetched once.
//              Then lane (plane q, row r): one row of the 36-tap centre patch.
//   section 2  lane (view slot, plane q): rows and anchors summed in the reference's order -> ev.
template <int SMP, int FMT>
DVP_HD void wave_ncc_new(const Dev& d, const PatchCtx& c, const s2* nbs, float cpix, int px, int py, uint32_t vmask, uint32_t pmask, WeakShared& sh) {
	const int W = d.width, Hh = d.height, Pt = d.pitch;
	const int np = 32 - __builtin_clz(pmask | 1u);   // plane slots in use (8 candidates, then 2 and 5)
	uint32_t rest = vmask;
	while (rest) {
		uint32_t batch = 0;
		for (int n = 0; n < kWeakPairs / np && n < kWeakViews && rest; ++n) { const uint32_t low = rest & (0u - rest); batch |= low; rest ^= low; }
		// Work items of the batch, 64 per round over the lanes:
		//   anchor items (view slot, anchor k, plane q), q fastest: the np planes of one (view, anchor) sit in
		//   adjacent lanes, so a load instruction's lanes share lines; a round is full whatever np is (the 2- and
		//   5-plane refinement phases and the second half of the 88 pairs of an 8-plane view used to leave
		//   60-75 % of a round's lanes idle);
		//   centre items (view slot, plane, patch row).
		// A lane's view changes from item to item, so the per-view constants are per-lane loads here.
		const int nv = __builtin_popcount(batch);
		const int rows_per = c.fast ? kTaps : 1;
		// the (view, anchor) pairs of the batch: what the anchor is worth in that view and its offsets, loaded by 64
		// lanes at a time BEFORE the items (every item used to start with two dependent loads — the anchor's
		// selected_views word, then its offsets — in front of its gathers)
		for (int i0 = 0; i0 < nv * kAnchors; i0 += 64) {
			DVP_LANES(l) {
				const int i = i0 + l;
				if (i >= nv * kAnchors) continue;
				const int slot = i / kAnchors, k = i - slot * kAnchors;
				const s2 nb = nbs[k + 1];
				int state = 0;
				if (!(nb.x == -1 || nb.y == -1)) {
					const int nbc = nb.x + nb.y * W;
					const int v0 = nth_set_bit(batch, slot);   // 0-based view
					state = is_set(d.selected_views[nbc], v0) ? 2 : 1;
					if (state == 2) {
						const s2* cand = d.candidate + cand_index(d, nbc, v0);
#pragma unroll
						for (int t = 0; t < 4; ++t) {
							const s2 a = cand[2 * t], b = cand[2 * t + 1];
							sh.aoff[i][t] = ((uint32_t)(uint8_t)(int8_t)a.x) | ((uint32_t)(uint8_t)(int8_t)a.y << 8) |
							                ((uint32_t)(uint8_t)(int8_t)b.x << 16) | ((uint32_t)(uint8_t)(int8_t)b.y << 24);
						}
					}
				}
				sh.astate[i] = (uint8_t)state;
			}
		}
		wave_sync();
		DVP_LANES(l) {
			const int n_anchor = nv * kAnchors * np;
			for (int it0 = 0; it0 < n_anchor; it0 += 64) {
				const int it = it0 + l;
				if (it >= n_anchor) continue;
				const int q = it % np, t2 = it / np;
				const int k = t2 % kAnchors, slot = t2 / kAnchors;
				if (!((pmask >> q) & 1)) continue;
				const int v = nth_set_bit(batch, slot) + 1;   // 1-based image index of the source view
				const ViewConst vc = d.views[v];
				float H[9];
				homography(vc, sh.pl[q], H);
				const f2 pt = apply_homography(H, px, py);
				const bool inside = !(pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f);
				if (k == 0) sh.inq[slot * np + q] = inside ? 1 : 0;
				if (!inside) continue;
				sh.acost[slot * np + q][k] = anchor_cost<SMP, FMT>(d, H, img_plane<FMT>(d, v), nbs[k + 1], sh.astate[slot * kAnchors + k], sh.aoff[slot * kAnchors + k], cpix);
			}
			const int n_centre = nv * np * rows_per;
			for (int it0 = 0; it0 < n_centre; it0 += 64) {
				const int it = it0 + l;
				if (it >= n_centre) continue;
				const int r = it % rows_per, t2 = it / rows_per;
				const int cq = t2 % np, slot = t2 / np;
				if (!((pmask >> cq) & 1)) continue;
				const int v = nth_set_bit(batch, slot) + 1;
				const ViewConst vc = d.views[v];
				float Hc[9];
				homography(vc, sh.pl[cq], Hc);
				const f2 pt = apply_homography(Hc, px, py);
				if (pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f) continue;
#if defined(DVP_ABL_NO_CENTRE)   // timing ablation (wrong results): what the launch costs without the centre patches
				continue;
#endif
				if (c.fast) {
					float o[3];
					patch_row_sums<SMP, FMT>(d, sh, Hc, img_plane<FMT>(d, v), px, py, c.radius, c.inc, r, o);
					sh.rows[slot * np + cq][r][0] = o[0];
					sh.rows[slot * np + cq][r][1] = o[1];
					sh.rows[slot * np + cq][r][2] = o[2];
				} else {   // generic (non-6-tap) patches sample the float planes
					sh.rows[slot * np + cq][0][0] = ncc_patch_generic(d, Hc, d.images + (size_t)v * d.plane_stride * 2, px, py, c.radius, c.inc, 1);
				}
			}
		}
		wave_sync();
		DVP_LANES(l) {
		for (int it0 = 0; it0 < nv * np; it0 += 64) {   // items (view slot, plane)
			const int it = it0 + l;
			if (it >= nv * np) continue;
			const int slot = it / np, q = it % np;
			if (!((pmask >> q) & 1)) continue;
			const int v = nth_set_bit(batch, slot);   // 0-based view index
			if (!sh.inq[slot * np + q]) { sh.ev[q][v] = 2.0f; continue; }
			float cc;
			if (c.fast) {
				float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
				for (int r = 0; r < kTaps; ++r) {
					s_s += sh.rows[slot * np + q][r][0];
					s_ss += sh.rows[slot * np + q][r][1];
					s_rs += sh.rows[slot * np + q][r][2];
				}
				cc = ncc_from_sums(c.sum_ref, c.sum_ref_ref, s_s, s_ss, s_rs, c.wsum);
			} else {
				cc = sh.rows[slot * np + q][0][0];
			}
			float scost = 0.0f, scnt = 0.0f;
			for (int k = 0; k < kAnchors; ++k) {
				const float ac = sh.acost[slot * np + q][k];
				if (ac >= 0.0f) { scost += ac; scnt += 1.0f; }
			}
			float out = cc;
			if (scnt > 0.0f) {
				float sc2 = scost / scnt;   // strong_cost /= strong_count (int -> float, exact)
				sc2 = DVP_MIN(sc2, 2.0f);
				out = (float)(0.25 * cc + 0.75 * sc2);
			}
			sh.ev[q][v] = out;
		}
		}
		wave_sync();
	}
}

// it / n for 0 <= it < 65536, 1 <= n <= 64 through one float multiplication (exact: (it + 0.5) / n is at least 1 / 2n away from
// an integer, the rounding error of the product is below 2^-7); the generic 32-bit division is ~25 instructions
DVP_HD int small_div(int it, float inv_n) { return (int)(((float)it + 0.5f) * inv_n); }

// wave_ncc_new with the anchor reference sides from the pass' table (TAB = 1).  Same evaluations, same bits; what changes
// is the bookkeeping around them:
//   * the homography of a (view, plane) pair is formed ONCE per batch into sh.Hq (section 0) instead of by each of the
//     pair's 17 items — 11 anchor sub-patches and 6 centre rows — (a quarter of an anchor round's instructions);
//   * the live planes are compacted: pair = view slot * np + (index among the set bits of pmask), np = popcount(pmask),
//     so the refinement phase's survivors (e.g. hypotheses 1 and 4 of 5) fill whole rounds;
//   * items decode with small_div.
template <int SMP, int FMT>
DVP_HD void wave_ncc_new_tab(const Dev& d, const PatchCtx& c, const s2* nbs, int px, int py, uint32_t vmask, uint32_t pmask, WeakSharedT<1>& sh) {
	const int W = d.width;
	const int np = __builtin_popcount(pmask);
	if (np == 0) return;
	const float inv_np = 1.0f / (float)np;
	const int weak_index = d.neighbours_map[px + py * W];
	uint32_t rest = vmask;
	while (rest) {
		uint32_t batch = 0;
		for (int n = 0; n < kWeakPairs / np && n < kWeakViews && rest; ++n) { const uint32_t low = rest & (0u - rest); batch |= low; rest ^= low; }
		const int nv = __builtin_popcount(batch);
		const int rows_per = c.fast ? kTaps : 1;
		// section 0: (view slot, anchor) states and the homographies of the (view slot, plane) pairs
		for (int i0 = 0; i0 < nv * kAnchors; i0 += 64) {
			DVP_LANES(l) {
				const int i = i0 + l;
				if (i >= nv * kAnchors) continue;
				const int slot = i / kAnchors, k = i - slot * kAnchors;
				const s2 nb = nbs[k + 1];
				int state = 0;
				if (!(nb.x == -1 || nb.y == -1)) state = is_set(d.selected_views[nb.x + nb.y * W], nth_set_bit(batch, slot)) ? 2 : 1;
				sh.astate[i] = (uint8_t)state;
			}
		}
		DVP_LANES(l) {
			if (l >= nv * np) continue;
			const int slot = small_div(l, inv_np), qi = l - slot * np;
			const int v = nth_set_bit(batch, slot) + 1;   // 1-based image index of the source view
			const ViewConst vc = d.views[v];
			float H[9];
			homography(vc, sh.pl[nth_set_bit(pmask, qi)], H);
			const f2 pt = apply_homography(H, px, py);
			sh.inq[l] = !(pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f) ? 1 : 0;
#pragma unroll
			for (int i = 0; i < 9; ++i) sh.Hq[l][i] = H[i];
		}
		wave_sync();
		// section 1: anchor items (view slot, anchor k, live plane qi), qi fastest — the np planes of one (view, anchor) sit in
		// adjacent lanes, so a load instruction's lanes share lines —, then centre items (pair, patch row)
		DVP_LANES(l) {
			const int n_anchor = nv * kAnchors * np;
			for (int it0 = 0; it0 < n_anchor; it0 += 64) {
				const int it = it0 + l;
				if (it >= n_anchor) continue;
				const int t2 = small_div(it, inv_np), qi = it - t2 * np;
				const int slot = t2 / kAnchors, k = t2 - slot * kAnchors;
				const int pair = slot * np + qi;
				if (!sh.inq[pair]) continue;
				const int v = nth_set_bit(batch, slot) + 1;
				float H[9];
#pragma unroll
				for (int i = 0; i < 9; ++i) H[i] = sh.Hq[pair][i];
				sh.acost[pair][k] = anchor_cost_tab<SMP, FMT>(d, H, img_plane<FMT>(d, v), nbs[k + 1], sh.astate[slot * kAnchors + k], d.anchor_tab + anchor_rec_index(d, weak_index, v - 1, k));
			}
			const int n_centre = nv * np * rows_per;
			for (int it0 = 0; it0 < n_centre; it0 += 64) {
				const int it = it0 + l;
				if (it >= n_centre) continue;
				const int pair = it / rows_per, r = it - pair * rows_per;
				if (!sh.inq[pair]) continue;
#if defined(DVP_ABL_NO_CENTRE)   // timing ablation (wrong results)
				continue;
#endif
				const int slot = small_div(pair, inv_np);
				const int v = nth_set_bit(batch, slot) + 1;
				float Hc[9];
#pragma unroll
				for (int i = 0; i < 9; ++i) Hc[i] = sh.Hq[pair][i];
				if (c.fast) {
					float o[3];
					patch_row_sums<SMP, FMT>(d, sh, Hc, img_plane<FMT>(d, v), px, py, c.radius, c.inc, r, o);
					sh.rows[pair][r][0] = o[0];
					sh.rows[pair][r][1] = o[1];
					sh.rows[pair][r][2] = o[2];
				} else {   // generic (non-6-tap) patches sample the float planes
					sh.rows[pair][0][0] = ncc_patch_generic(d, Hc, d.images + (size_t)v * d.plane_stride * 2, px, py, c.radius, c.inc, 1);
				}
			}
		}
		wave_sync();
		// section 2: lane = pair: rows and anchors summed in the reference's order -> ev
		DVP_LANES(l) {
			for (int it0 = 0; it0 < nv * np; it0 += 64) {
				const int pair = it0 + l;
				if (pair >= nv * np) continue;
				const int slot = small_div(pair, inv_np), qi = pair - slot * np;
				const int q = nth_set_bit(pmask, qi);
				const int v = nth_set_bit(batch, slot);   // 0-based view index
				if (!sh.inq[pair]) { sh.ev[q][v] = 2.0f; continue; }
				float cc;
				if (c.fast) {
					float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
					for (int r = 0; r < kTaps; ++r) {
						s_s += sh.rows[pair][r][0];
						s_ss += sh.rows[pair][r][1];
						s_rs += sh.rows[pair][r][2];
					}
					cc = ncc_from_sums(c.sum_ref, c.sum_ref_ref, s_s, s_ss, s_rs, c.wsum);
				} else {
					cc = sh.rows[pair][0][0];
				}
				float scost = 0.0f, scnt = 0.0f;
				for (int k = 0; k < kAnchors; ++k) {
					const float ac = sh.acost[pair][k];
					if (ac >= 0.0f) { scost += ac; scnt += 1.0f; }
				}
				float out = cc;
				if (scnt > 0.0f) {
					float sc2 = scost / scnt;   // strong_cost /= strong_count (int -> float, exact)
					sc2 = DVP_MIN(sc2, 2.0f);
					out = (float)(0.25 * cc + 0.75 * sc2);
				}
				sh.ev[q][v] = out;
			}
		}
		wave_sync();
	}
}

// gtab[q][j] = ComputeGeomConsistencyCost(pixel, view j+1, sh.pl[q]) for q in pmask, views with weight > 0
template <class SH>
DVP_HD void wave_geom_table(const Dev& d, const DvpCamera& rc, int px, int py, uint32_t pmask, SH& sh) {
	const int S = d.params.num_images - 1;
	for (int j0 = 0; j0 < S; j0 += 8) {
		DVP_LANES(l) {
			const int q = l >> 3, j = j0 + (l & 7);
			if (j >= S || !((pmask >> q) & 1) || sh.vw[j] == 0) continue;
			sh.gtab[q][j] = geom_cost_cams(d, rc, d.cameras[j + 1], j + 1, px, py, sh.pl[q]);
		}
	}
	wave_sync();
}

template <int SMP, int FMT>
DVP_HD void weak_eval(const Dev& d, const PatchCtx& c, const s2* nbs, float cpix, int px, int py, uint32_t vmask, uint32_t pmask, WeakSharedT<0>& sh) {
	wave_ncc_new<SMP, FMT>(d, c, nbs, cpix, px, py, vmask, pmask, sh);
}
template <int SMP, int FMT>
DVP_HD void weak_eval(const Dev& d, const PatchCtx& c, const s2* nbs, float cpix, int px, int py, uint32_t vmask, uint32_t pmask, WeakSharedT<1>& sh) {
	wave_ncc_new_tab<SMP, FMT>(d, c, nbs, px, py, vmask, pmask, sh);
}

template <int SMP, int FMT, int TAB>
DVP_HD void weak_update_wave(const Dev& d, int px, int py, int iter, unsigned long long* nevals, WeakSharedT<TAB>& sh) {
	const int W = d.width, Hh = d.height;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const uint32_t all_views = (S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);
	const s2* nbs = d.neighbours + (size_t)d.neighbours_map[center] * DVP_NEIGHBOUR_NUM;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	unsigned long long evals = 0;

	PatchCtx c;
	c.tab = PatchTab{nullptr, 0};
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		wave_patch_ctx<FMT>(d, px, py, radius, inc, 1, sh, &c);
	}
	DVP_LANES(l) {
		if (l < 32) sh.vw[l] = 0;
		if (l < 8) sh.positions[l] = 0;
	}
	wave_sync();
	// (cost_array: `= { 2.0f }` sets one element, the rest is 0 (APD.cu:2769) — written whole by the epilogue of the propagation
	// phase below; until then its bytes hold the batch homographies of the TAB form)
	uint32_t flag = 0;
	uint32_t sel_mask = 0;
	uint32_t sel_now = d.selected_views[center];   // what random_normal_yzl reads (updated on adoption)
	float weight_norm = 0.0f;
	int min_cost_idx = 0;
	float cost_now = 0.0f, costs_center = 0.0f, depth_now = 0.0f;
	f4 plane_now = mk4(0, 0, 0, 0);
	bool skip_refine = false;
	f4 pl1 = mk4(0, 0, 0, 0);

	for (int phase = 0; phase < 3; ++phase) {
		uint32_t pmask = 0, vmask = 0;
		if (phase == 0) {
			for (int k = 0; k < 8; ++k) {
				const s2 nb = nbs[k + 1];
				if (!(nb.x == -1 || nb.y == -1) && d.weak_info[nb.x + nb.y * W] == DVP_STRONG) {
					flag |= 1u << k;
					if (DVP_LANE0) {
						sh.positions[k] = nb.x + nb.y * W;
						sh.pl[k] = d.planes[nb.x + nb.y * W];
					}
				}
			}
			pmask = flag;
			vmask = all_views;
		} else if (phase == 1) {
			// joint view selection (APD.cu:2781-2850): lane j owns view j for the per-view parts
			const float thr = (float)(0.8 * dvp_expf((iter) * (iter) / (-90.0f)));
			DVP_LANES(j) {
				if (j >= S) continue;
				float pr = 0.0f;
				for (int i = 0; i < 8; ++i) {
					const s2 nb = nbs[i + 1];
					if (nb.x == -1 || nb.y == -1) continue;
					pr += is_set(d.selected_views[nb.x + nb.y * W], j) ? 0.9f : 0.1f;
				}
				float count = 0;
				int count_false = 0;
				float tmpw = 0;
				for (int k = 0; k < 8; k++) {
					const float cst = sh.cost_array[k][j];
					if (cst < thr) { tmpw += dvp_expf(cst * cst / (-0.18f)); count++; }
					if (cst > 1.2f) count_false++;
				}
				float p = 0.0f;
				if (count > 2 && count_false < 3) p = tmpw / count;
				else if (count_false < 3) p = dvp_expf(thr * thr / (-0.32f));
				sh.probs[j] = p * pr;
			}
			wave_sync();
			float psum = 0.0f;
			for (int i = 0; i < S; ++i) psum += sh.probs[i];
			const float inv = 1.0f / psum;
			float cum = 0.0f;
			wave_sync();
			for (int i = 0; i < S; ++i) {
				cum += sh.probs[i] * inv;
				if (DVP_LANE0) sh.priors[i] = cum;   // the CDF (priors[] is free from here on)
			}
			wave_sync();
			Rng rv(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_VIEW));
			for (int s = 0; s < 15; ++s) {
				const float rp = rv.uniform() - FLT_EPSILON;
				for (int v = 0; v < S; ++v)
					if (sh.priors[v] > rp) { if (DVP_LANE0) sh.vw[v] += 1; break; }
			}
			wave_sync();
			sel_mask = 0;
			weight_norm = 0;
			for (int i = 0; i < S; ++i)
				if (sh.vw[i] > 0) { set_bit(&sel_mask, i); weight_norm += sh.vw[i]; }
			DVP_LANES(i) { if (i < 32) d.view_weight[(size_t)center * 32 + i] = sh.vw[i]; }
			// weighted candidate costs (APD.cu:2852-2874); sh.pl[k] still holds the anchors' planes
			if (P.geom_consistency) wave_geom_table(d, rc, px, py, flag, sh);
			DVP_LANES(k) {
				if (k >= 8) continue;
				float fc = 0.0f;
				for (int j = 0; j < S; ++j) {
					const int w = sh.vw[j];
					if (w > 0) {
						if (P.geom_consistency) {
							if ((flag >> k) & 1) fc += w * (sh.cost_array[k][j] + P.geom_factor * sh.gtab[k][j]);
							else fc += w * (sh.cost_array[k][j] + P.geom_factor * 3.0f);
						} else {
							fc += w * sh.cost_array[k][j];
						}
					}
				}
				sh.fcost[k] = fc / weight_norm;
			}
			wave_sync();
			min_cost_idx = 0;
			{
				float mc = sh.fcost[0];
				for (int k = 1; k < 8; ++k)
					if (sh.fcost[k] <= mc) { mc = sh.fcost[k]; min_cost_idx = k; }
			}
			const f4 fp = d.fit_planes[center];
			pl1 = fp;
			pmask = 1u;
			if (fp.x == 0 && fp.y == 0 && fp.z == 0) skip_refine = true;
			else pmask |= 2u;
			wave_sync();
			if (DVP_LANE0) {
				sh.pl[0] = d.planes[center];
				if (!skip_refine) sh.pl[1] = fp;
			}
			vmask = sel_mask;
		} else {
			if (skip_refine) break;
			Rng rd(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_DEPTH_RAND));
			Rng rn(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_NORMAL));
			Rng rp(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_DEPTH_PERT));
			const float depth_rand = rd.uniform() * (P.depth_max - P.depth_min) + P.depth_min;
			const f4 n_rand = random_normal_yzl_sel(d, px, py, rn, depth_now, sel_now);
			const float dmin_p = (1 - 0.02f) * depth_now, dmax_p = (1 + 0.02f) * depth_now;
			const float depth_pert = rp.uniform() * (dmax_p - dmin_p) + dmin_p;
			f4 n_pert = plane_now;
			normalize3(&n_pert);
			const float rdep[5] = { depth_rand, depth_now, depth_rand, depth_now, depth_pert };
			const f4 rnrm[5] = { plane_now, n_rand, n_rand, n_pert, plane_now };
			wave_sync();
#pragma unroll
			for (int i = 0; i < 5; ++i) {
				f4 h = rnrm[i];
				h.w = distance_to_origin(rc, px, py, rdep[i], h);
				if (DVP_LANE0) sh.pl[i] = h;
			}
			pmask = 0x1Fu;
			vmask = sel_mask;
		}
		wave_sync();

		// ---- evaluate: view by view ----------------------------------------------------------------------
		if (phase == 2 && pmask && vmask) {
			// The five refinement hypotheses are adopted iff their depth is in range and their weighted cost is below the
			// running best, which only ever decreases (APD.cu:1361-1383).  Two exact short cuts — the costs themselves are
			// only kept on adoption: (i) out-of-range hypotheses are not evaluated; (ii) the weighted sum only grows (weights
			// > 0, costs >= 0, IEEE addition and division are monotone), so a hypothesis whose FIRST selected view alone is
			// not below the best cost at entry can never be adopted: it drops out before the remaining views are evaluated.
			uint32_t keep = 0;
			for (int i = 0; i < 5; ++i) {
				const float db = depth_from_plane(rc, sh.pl[i], px, py);
				if (db >= P.depth_min && db <= P.depth_max) keep |= 1u << i;
			}
			pmask = keep;
			int first = 0;
			while (!((vmask >> first) & 1)) ++first;
			if (pmask) {
				weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, 1u << first, pmask, sh);
				evals += (unsigned long long)__builtin_popcount(pmask);
				if (P.geom_consistency) wave_geom_table(d, rc, px, py, pmask, sh);
				uint32_t alive = 0;
				const int w = sh.vw[first];
				for (int i = 0; i < 5; ++i) {
					if (!((pmask >> i) & 1)) continue;
					float tc = 0.0f;
					if (P.geom_consistency) tc += w * (sh.ev[i][first] + P.geom_factor * sh.gtab[i][first]);
					else tc += w * sh.ev[i][first];
					if (tc / weight_norm < cost_now) alive |= 1u << i;
				}
				pmask = alive;
				const uint32_t rest = vmask & ~(1u << first);
				if (pmask && rest) {
					weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, rest, pmask, sh);
					evals += (unsigned long long)__builtin_popcount(pmask) * (unsigned long long)__builtin_popcount(rest);
				}
			}
		} else if (pmask && vmask) {
			weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, vmask, pmask, sh);
			evals += (unsigned long long)__builtin_popcount(pmask) * (unsigned long long)__builtin_popcount(vmask);
		}

		// ---- epilogue ----------------------------------------------------------------------------------------
		if (phase == 0) {
			DVP_LANES(l) {
				for (int i = l; i < 8 * 32; i += 64) {
					const int k = i >> 5, v = i & 31;
					sh.cost_array[k][v] = (((flag >> k) & 1) && v < S) ? sh.ev[k][v] : (i == 0 ? 2.0f : 0.0f);
				}
			}
			wave_sync();
		} else if (phase == 1) {
			if (P.geom_consistency) wave_geom_table(d, rc, px, py, pmask, sh);
			float cn = 0.0f;
			for (int v = 0; v < S; ++v) {
				const int w = sh.vw[v];
				if (w > 0) {
					if (P.geom_consistency) cn += w * (sh.ev[0][v] + P.geom_factor * sh.gtab[0][v]);
					else cn += w * sh.ev[0][v];
				}
			}
			cost_now = cn / weight_norm;
			costs_center = cost_now;
			plane_now = sh.pl[0];
			depth_now = depth_from_plane(rc, plane_now, px, py);
			if ((flag >> min_cost_idx) & 1) {
				const f4 cand = d.planes[sh.positions[min_cost_idx]];
				const float db = depth_from_plane(rc, cand, px, py);
				if (db >= P.depth_min && db <= P.depth_max && sh.fcost[min_cost_idx] < cost_now) {
					depth_now = db;
					plane_now = cand;
					cost_now = sh.fcost[min_cost_idx];
					sel_now = sel_mask;
					if (DVP_LANE0) d.selected_views[center] = sel_mask;
				}
			}
			if (!skip_refine) {   // fit-plane test
				float tc = 0.0f;
				for (int j = 0; j < S; ++j) {
					const int w = sh.vw[j];
					if (w > 0) {
						if (P.geom_consistency) tc += w * (sh.ev[1][j] + P.geom_factor * sh.gtab[1][j]);
						else tc += w * sh.ev[1][j];
					}
				}
				tc /= weight_norm;
				const float db = depth_from_plane(rc, pl1, px, py);
				if (db >= P.depth_min && db <= P.depth_max && tc < cost_now) {
					depth_now = db;
					plane_now = pl1;
					cost_now = tc;
				}
			}
		} else {
			// (geometric table: filled in the evaluate step; pmask = the hypotheses that are still candidates)
			for (int i = 0; i < 5; ++i) {
				if (!((pmask >> i) & 1)) continue;
				float tc = 0.0f;
				for (int j = 0; j < S; ++j) {
					const int w = sh.vw[j];
					if (w > 0) {
						if (P.geom_consistency) tc += w * (sh.ev[i][j] + P.geom_factor * sh.gtab[i][j]);
						else tc += w * sh.ev[i][j];
					}
				}
				tc /= weight_norm;
				const f4 h = sh.pl[i];
				const float db = depth_from_plane(rc, h, px, py);
				if (db >= P.depth_min && db <= P.depth_max && tc < cost_now) {
					depth_now = db;
					plane_now = h;
					cost_now = tc;
				}
			}
		}
	}

	f4 final_plane = d.planes[center];
	if (P.state == DVP_REFINE_INIT) {
		if (cost_now < costs_center - 0.1) final_plane = plane_now;
	} else {
		final_plane = plane_now;
	}
	wave_sync();
	if (DVP_LANE0) d.planes[center] = final_plane;

	// cost of the final plane with the plain bilateral NCC at the default radius (APD.cu:3072-3088):
	// lane = (view slot, patch row), eight views per round
	PatchCtx c2;
	c2.tab = PatchTab{nullptr, 0};
	{
		int r = P.strong_radius, inc = P.strong_increment;
		if (P.use_radius) inc = DVP_MAX(2, (int)(2.0 * r / 5.0));
		wave_patch_ctx<FMT>(d, px, py, r, inc, 0, sh, &c2);
	}
	for (int v0 = 0; v0 < S; v0 += 8) {
		DVP_LANES(l) {
			const int vs = l >> 3, r = l & 7, v = v0 + vs;
			if (v >= S || sh.vw[v] == 0) continue;
			if (c2.fast ? r >= kTaps : r != 0) continue;
			const ViewConst vc = d.views[v + 1];
			float H[9];
			homography(vc, final_plane, H);
			const f2 pt = apply_homography(H, px, py);
			const bool in = !(pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f);
			const void* src = img_plane<FMT>(d, v + 1);
			if (!c2.fast) {
				sh.ev[0][v] = in ? ncc_patch_generic(d, H, d.images + (size_t)(v + 1) * d.plane_stride * 2, px, py, c2.radius, c2.inc, 0) : 2.0f;
			} else if (in) {
				float o[3];
				patch_row_sums<SMP, FMT>(d, sh, H, src, px, py, c2.radius, c2.inc, r, o);
				sh.rows[vs][r][0] = o[0];
				sh.rows[vs][r][1] = o[1];
				sh.rows[vs][r][2] = o[2];
			} else if (r == 0) {
				sh.ev[0][v] = 2.0f;
				sh.inq[vs] = 0;   // "outside" for the totalling lane
			}
			if (c2.fast && in && r == 0) sh.inq[vs] = 1;
		}
		wave_sync();
		if (c2.fast) {
			DVP_LANES(l) {
				const int v = v0 + l;
				if (l >= 8 || v >= S || sh.vw[v] == 0 || !sh.inq[l]) continue;
				float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
				for (int r = 0; r < kTaps; ++r) {
					s_s += sh.rows[l][r][0];
					s_ss += sh.rows[l][r][1];
					s_rs += sh.rows[l][r][2];
				}
				sh.ev[0][v] = ncc_from_sums(c2.sum_ref, c2.sum_ref_ref, s_s, s_ss, s_rs, c2.wsum);
			}
			wave_sync();
		}
	}
	float cn = 0.0f;
	for (int v = 0; v < S; ++v) {
		if (sh.vw[v] == 0) continue;
		cn += sh.vw[v] * sh.ev[0][v];
		evals += 1;
	}
	if (DVP_LANE0) {
		d.costs[center] = cn / weight_norm;
		if (nevals) *nevals += evals;
	}
	wave_sync();
}

// ---- GenNeighbours, first half (APD.cu:3330-3505): the anchor search, ONE WAVE PER WEAK PIXEL ------------------------
// gen_neighbours_px (dvp_weak.hpp) walks, per lane, 32 directions x a sequence of tries with two to four dependent loads
// each (and a line walk per surviving try): 80 ms per cfg3 pass at 7 % VALU activity, the slowest lane of 64 unrelated
// pixels setting the pace of every direction.  What is sequential in it is only (i) the shared random stream — but a try
// always consumes exactly four draws, so try t of a direction that starts at counter k0 uses k0 + 4 t .. + 3 — and (ii)
// "this point was already found by an earlier direction".  Within ONE direction the tries are therefore independent
// functions of (k0, t): the lanes evaluate tries t0 .. t0 + kGnTries - 1 at once (RNG, candidate point, STRONG test,
// nearest-strong fall-back, duplicate test against the earlier directions, angle test), the edge-limited line walks of
// the surviving tries run eight tries at a time, and the direction takes the FIRST try that passes, exactly like the
// sequential loop; the stream advances by 4 x (tries the sequential loop would have made).  The label-extension points
// are found the same way ((direction, step) items over the lanes) and appended in item order.  Same points, same order,
// same random numbers: the per-lane function stays in the tree as the definition and the tests compare the two.
#ifndef DVP_GN_TRIES
#define DVP_GN_TRIES 64
#endif
constexpr int kGnTries = DVP_GN_TRIES;   // tries of one direction evaluated per round (<= 64)
struct GnShared {
	s2 pts[kGnDirSlots];    // the directional slots (holes = (-1,-1))
	s2 cnp[64];             // the candidate of this lane's try
	uint8_t flag[64];       // 0 = try failed, 1 = try accepted, 2 = the try does not exist, 4 = candidate, line walk pending
};
// first lane < limit whose flag has one of the bits in `mask` (64 = none)
DVP_HD int wave_first_flag(const uint8_t* flag, int mask, int limit) {
#if defined(__HIP_DEVICE_COMPILE__)
	const int l = (int)(threadIdx.x & 63u);
	const unsigned long long m = __ballot(l < limit && (flag[l] & mask) != 0);
	return m ? (int)__builtin_ctzll(m) : 64;
#else
	for (int l = 0; l < limit; ++l) if (flag[l] & mask) return l;
	return 64;
#endif
}
#if defined(DVP_GN_STATS) && !defined(__HIPCC__)
// test instrumentation (tests/emul, -DDVP_GN_STATS): histogram of the try index a direction ends at
struct GnStats { long hist[8] = {0}, found = 0, out = 0; ~GnStats() { fprintf(stderr, "gn tries <1 <4 <8 <16 <32 <64 <128 more: %ld %ld %ld %ld %ld %ld %ld %ld found %ld out %ld\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], found, out); } };
inline void gn_stats_note(int t, bool found) {
	static GnStats g;
	const int lim[7] = { 1, 4, 8, 16, 32, 64, 128 };
	int b = 7;
	for (int i = 6; i >= 0; --i) if (t < lim[i]) b = i;
#pragma omp critical
	{ g.hist[b]++; if (found) g.found++; else g.out++; }
}
#endif
DVP_HD int gn_radius_of_group(int g) { return g <= 4 ? (2 << g) : 32 + 25 * (g - 4); }   // 2, 4, 8, 16, 32, 57, 82, ... (APD.cu:3449: min(2r, r + 25))

DVP_HD void gen_neighbours_search_wave(const Dev& d, int px, int py, GnShared& sh) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const int min_margin = 6;
	const int wi = d.neighbours_map[center];
	s2* neighbours = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	const uint32_t site_search = rng_site(PH_NEIGHBOURS, 0, SUB_SEARCH);
	DVP_LANES(l) {
		if (l < DVP_NEIGHBOUR_NUM) neighbours[l] = l == 0 ? mks2(px, py) : mks2(-1, -1);
		if (l < 32) sh.pts[l] = mks2(-1, -1);
	}
	wave_sync();
	int strong_point_size = 0;
	const int rotate_time = P.rotate_time;
	bool edge_limit = false;
	if (P.use_limit) {
		edge_limit = true;
		if (P.use_edge) {
			Rng r_limit(d.seed, (uint32_t)center, rng_site(PH_NEIGHBOURS, 0, SUB_LIMIT));
			const float rp = r_limit.uniform() - FLT_EPSILON;
			if (rp < d.complex_[wi]) edge_limit = false;
		}
	}
	uint32_t k0 = 0;   // r_search's counter at the start of the current direction
	int odi = -1;
	for (int odx = -1; odx <= 1; ++odx) {
		for (int ody = -1; ody <= 1; ++ody) {
			if (odx == 0 && ody == 0) continue;
			f2 od = mk2((float)odx, (float)ody);
			normalize2(&od);
			odi++;
			for (int rot = 0; rot < rotate_time; ++rot) {
				const int dir_index = odi * 4 + rot;
				bool resolved = false;
				for (int t0 = 0; !resolved; t0 += kGnTries) {
					DVP_LANES(l) {
						const int t = t0 + l;
						const int cur_radius = gn_radius_of_group(t >> 2);
						uint8_t f = 0;
						s2 np = mks2(-1, -1);
						// the try exists iff its radius group was entered: radius <= 4096 and the ray still inside the image
						const float tx = px + od.x * cur_radius, ty = py + od.y * cur_radius;
						if (l >= kGnTries) f = 0;
						else if (cur_radius > 4096 || tx < 0 || ty < 0 || tx >= W || ty >= H) f = 2;
						else {
							const uint32_t kk = k0 + 4u * (uint32_t)t;
							const uint32_t sgx = (rand_u32(d.seed, (uint32_t)center, site_search, kk) % 2 == 0) ? 1u : 0xFFFFFFFFu;
							const int xs = (int)((sgx * rand_u32(d.seed, (uint32_t)center, site_search, kk + 1)) % (uint32_t)d.nb_shift_range);
							const uint32_t sgy = (rand_u32(d.seed, (uint32_t)center, site_search, kk + 2) % 2 == 0) ? 1u : 0xFFFFFFFFu;
							const int ys = (int)((sgy * rand_u32(d.seed, (uint32_t)center, site_search, kk + 3)) % (uint32_t)d.nb_shift_range);
							f2 dir = mk2(od.x * 20 + xs, od.y * 20 + ys);
							normalize2(&dir);
							np = mks2((int)(px + dir.x * cur_radius), (int)(py + dir.y * cur_radius));
							bool ok = !(np.x < min_margin || np.y < min_margin || np.x >= W - min_margin || np.y >= H - min_margin);
							if (ok && !strong_bit(d, np.x, np.y)) {
								np = d.weak_nearest_strong[np.x + np.y * W];
								ok = !(np.x == -1 || np.y == -1);
							}
							if (ok) {
								bool same = false;
								for (int k = 0; k < dir_index; k++) same |= (sh.pts[k].x == np.x) & (sh.pts[k].y == np.y);
								f2 td = mk2((float)(np.x - px), (float)(np.y - py));
								normalize2(&td);
								const float cos_a = td.x * od.x + td.y * od.y;
								if (!same && cos_a > d.nb_thresh) f = edge_limit ? 4 : 1;
							}
						}
						sh.cnp[l] = np;
						sh.flag[l] = f;
					}
					wave_sync();
					// the sequential loop's outcome: the first try that is accepted (1) or does not exist (2); a pending try (4)
					// in front of it has to be walked first — eight tries per round, most directions end in the first
					int first = 64;
					for (int r0 = 0; r0 < kGnTries; r0 += 8) {
						if (edge_limit) {
							DVP_LANES(l) {
								if (l >= r0 && l < r0 + 8 && sh.flag[l] == 4) {
									const s2 np = sh.cnp[l];
									sh.flag[l] = bresenham_hits_edge(d, px, py, np.x, np.y) ? 0 : 1;
								}
							}
							wave_sync();
						}
						first = wave_first_flag(sh.flag, 3, r0 + 8);
						if (first < 64) break;
					}
#if defined(DVP_GN_STATS) && !defined(__HIPCC__)
					if (first < 64) gn_stats_note(t0 + first, sh.flag[first] == 1);
#endif
					if (first < 64) {
						resolved = true;
						if (sh.flag[first] == 1) {
							const s2 np = sh.cnp[first];
							wave_sync();
							if (DVP_LANE0) sh.pts[dir_index] = np;
							strong_point_size++;
							k0 += 4u * (uint32_t)(t0 + first + 1);
						} else {
							k0 += 4u * (uint32_t)(t0 + first);
						}
					}
					wave_sync();
				}
				f2 rd;
				rd.x = od.x * d.nb_cos - od.y * d.nb_sin;
				rd.y = od.x * d.nb_sin + od.y * d.nb_cos;
				normalize2(&rd);
				od = rd;
			}
		}
	}

	// hand-over: the 32 directional slots with their holes, and how many are filled (the same as gen_neighbours_px)
	s2* out = d.gn_points + (size_t)wi * kGnDirSlots;
	DVP_LANES(l) { if (l < kGnDirSlots) out[l] = sh.pts[l]; }
	if (DVP_LANE0) d.gn_count[wi] = strong_point_size;
}

// ---- GenNeighbours, the label extension (APD.cu:3455-3560): one wave per WEAK pixel --------------------------------
// Points on 16 rays through the pixel's label region, every one mapped to a STRONG pixel and appended to the list unless
// it is in the list already.  The (ray, step) items are independent but for that duplicate test: 64 items per round over
// the lanes, each tested against the list as it stood before the round and against the EARLIER items of the round, then
// appended in item order — the list the reference's sequential loop builds.  `pts` is the list (LDS, kGnMaxPoints
// entries, the first 32 filled by the directional search); returns the index of the last entry, *n_appended how many
// were added.  cnp / flag / keep: 64 entries of shared scratch each.
DVP_HD int gen_neighbours_extend_wave(const Dev& d, int px, int py, int wi, s2* pts, s2* cnp, uint8_t* flag, uint8_t* keep, int* n_appended) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const int min_margin = 6;
	const int rotate_time = P.rotate_time;
	int extend_index = kGnDirSlots - 1;
	int total_appended = 0;
	if (P.use_label && d.label[center] > 0) {
		const int ldx[16] = { 0, 0, -1, 1, -1, 1, -1, 1, 1, 0, 0, -1, -1, 0, 0, 1 };   // APD.cu:3462 (0.5 -> 0)
		const int ldy[16] = { -1, 1, 0, 0, -1, 1, 1, -1, 0, 1, 1, 0, 0, -1, -1, 0 };
		const s2* lb = d.label_boundary + (size_t)wi * 8;
		float bound_dist[16];
		int dir_step[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) { bound_dist[i] = 0.0f; dir_step[i] = 0; }
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const s2 bp = lb[i];
			float dist = 0.0f;
			if (bp.x != -1 && bp.y != -1) {
				const double ex = (double)(px - bp.x), ey = (double)(py - bp.y);
				dist = (float)sqrt(ex * ex + ey * ey);
				if (i >= 4) dist = (float)((double)dist / sqrt(2.0));
			}
			bound_dist[i] = dist;
			if (i % 2 == 1) { dir_step[i - 1] = 4 * rotate_time - 1; dir_step[i] = 1; }   // APD.cu:3477: step == 1
		}
		const int ca[8] = { 3, 1, 1, 2, 2, 4, 7, 7 };
		const int cb[8] = { 5, 5, 6, 6, 4, 0, 0, 3 };
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			dir_step[8 + q] = (dir_step[ca[q]] + dir_step[cb[q]]) / 2;
			bound_dist[8 + q] = (bound_dist[ca[q]] + bound_dist[cb[q]]) / 2;
		}
		int n_items = 0;     // items = (direction i, step 1 .. dir_step[i]) in the reference's loop order
#pragma unroll
		for (int i = 0; i < 16; ++i) n_items += dir_step[i] > 0 ? dir_step[i] : 0;
		for (int c0 = 0; c0 < n_items; c0 += 64) {
			DVP_LANES(l) {
				const int c = c0 + l;
				uint8_t f = 0;
				s2 np = mks2(-1, -1);
				if (c < n_items) {
					int first = 0, step = 0, step_len = 1, ddx = 0, ddy = 0;   // (a select chain: the tables stay in registers)
#pragma unroll
					for (int i = 0; i < 16; ++i) {
						const int n = dir_step[i] > 0 ? dir_step[i] : 0;
						if (c >= first && c < first + n) {
							step = c - first + 1;
							step_len = DVP_MAX(1, (int)floor(1.0 * bound_dist[i] / (dir_step[i] + 1)));
							ddx = ldx[i];
							ddy = ldy[i];
						}
						first += n;
					}
					np = mks2(px + step * step_len * ddx, py + step * step_len * ddy);
					bool ok = !(np.x < min_margin || np.y < min_margin || np.x >= W - min_margin || np.y >= H - min_margin);
					if (ok && !strong_bit(d, np.x, np.y)) {
						np = d.weak_nearest_strong[np.x + np.y * W];
						ok = !(np.x == -1 || np.y == -1);
					}
					if (ok) {
						bool same = false;      // against everything accepted before this batch ...
						for (int k = 0; k <= extend_index; k++) same |= (pts[k].x == np.x) & (pts[k].y == np.y);
						if (!same) f = 1;
					}
				}
				cnp[l] = np;
				flag[l] = f;
			}
			wave_sync();
			DVP_LANES(l) {
				// ... and against the earlier items of the batch: the first occurrence of a point is the one the sequential
				// loop appends (an earlier equal item that was itself dropped had an even earlier equal one)
				const s2 np = cnp[l];
				bool kp = flag[l] == 1;
				for (int q = 0; q < l; ++q) kp &= !((flag[q] == 1) & (cnp[q].x == np.x) & (cnp[q].y == np.y));
				keep[l] = kp ? 1 : 0;
			}
			wave_sync();
			int appended = 0;
			DVP_LANES(l) {
				int before = 0, total = 0;    // kept items in front of this one / in the batch
#if defined(__HIP_DEVICE_COMPILE__)
				{
					const unsigned long long m = __ballot(keep[l] != 0);
					total = __popcll(m);
					before = __popcll(m & ((1ull << l) - 1ull));
				}
#else
				for (int k = 0; k < 64; ++k) { total += keep[k]; before += k < l ? keep[k] : 0; }
#endif
				const int pos = extend_index + 1 + before;
				if (keep[l] && pos < kGnMaxPoints) pts[pos] = cnp[l];
				appended = DVP_MIN(total, kGnMaxPoints - 1 - extend_index);
			}
			wave_sync();
			extend_index += appended;
			total_appended += appended;
		}
	}
	*n_appended = total_appended;
	return extend_index;
}

// ---- GenNeighbours, second half (APD.cu:3562-3711): RANSAC plane through the candidate anchors, then the
// anchors ranked by their distance to that plane — ONE WAVE per WEAK pixel.
//
// A pixel per lane keeps its 160-entry point tables in scratch and reads them at random indices for 200
// draws: 64 lanes x divergent indices = one cache sector per access, 600 GB of fetches per launch at
// 6208x4128 with the vector ALUs 28 % busy.  Here the tables of one pixel live in LDS and the wave splits
// the work that is independent:
//   1. the 200 draws (index triples + the cheap rejections) — the RNG is counter-based and every draw
//      consumes exactly three numbers, so draw t owns counters 3t..3t+2;
//   2. in draw order (cheap, every lane in step): which unordered pairs need a line test and in which
//      orientation they are asked FIRST — the reference caches the test per pair, symmetric, first asker
//      wins (APD.cu:3574, 3588-3604);
//   3. the line walks, one per lane;
//   4. per draw: plane, label test, inlier count, distance of the pixel's own depth to the plane;
//   5. in draw order (every lane in step): the reference's running best (APD.cu:3646-3668);
//   6. residuals of all points, their stable rank by residual (== the reference's insertion sort,
//      APD.cu:125-138), neighbours[1..11].
constexpr int kGnDraws = 200;
constexpr int kGnPairWords = (kGnMaxPoints * (kGnMaxPoints - 1) / 2 + 31) / 32;
struct FitShared {
	uint8_t slot_of[kGnMaxPoints]; // valid point j -> index in raw
	s2 spv[kGnMaxPoints];
	f3 sp3[kGnMaxPoints];          // camera-frame 3-D point
	f2 fxy[kGnMaxPoints];          // ((x - cx) / fx, (y - cy) / fy)
	union { s2 raw[kGnMaxPoints]; float weight[kGnMaxPoints]; };   // the list as handed over (holes = (-1,-1)), dead once the tables are built; then the residuals
	uint32_t trip[kGnDraws];       // a | b << 8 | c << 16 | passed << 24
	uint32_t seen[kGnPairWords];   // unordered pair already queued for its line test
	uint32_t hit[kGnPairWords];    // ... and the test found an edge pixel
	uint16_t walk[3 * kGnDraws];   // queued tests: from | to << 8 (orientation of the first asker)
	float cand_dist[kGnDraws];
	int cand_info[kGnDraws];       // bit 0 valid, bit 1 "strong plane", bits 8..: inlier count
	uint8_t plist[kGnDraws];       // draws that passed the index and triangle tests, in draw order
	int req_idx[64], mark[64];     // one batch of line-test requests: pair index / pair to mark
	uint8_t req_from[64], req_to[64];
	int part_t[64];                // per-lane winner of the running best
	uint8_t part_nan[64];
	int n_pass, n_walk, n_valid;
};

// shared-memory counters / bit sets touched by several lanes of the wave in one section
DVP_HD int wave_counter_add(int* counter) {
#if defined(__HIP_DEVICE_COMPILE__)
	return atomicAdd(counter, 1);
#else
	return (*counter)++;
#endif
}
// Slot of the calling lane in a list that the lanes with `pred` append to IN LANE ORDER; *counter (shared memory) is the
// list length.  Every lane of the wave must call it (no early exit before it inside the section).
DVP_HD int wave_ordered_slot(bool pred, int* counter) {
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned long long m = __ballot(pred);
	const int base = *counter;
	const unsigned lane = threadIdx.x & 63u;
	wave_sync();
	if (lane == 0) *counter = base + __popcll(m);
	return base + __popcll(m & ((1ull << lane) - 1ull));
#else
	const int r = *counter;
	if (pred) (*counter)++;
	return r;
#endif
}
DVP_HD void wave_bits_or(uint32_t* word, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
	atomicOr(word, bits);
#else
	*word |= bits;
#endif
}

// The RANSAC inlier test is `|fit_depth - z| / depth_diff < ransac_threshold` (APD.cu:3632) — one IEEE division per (candidate
// plane, point) in the hottest loop of the launch (44 % of dvp_gen_neighbours_fit at 25 % WEAK, tools/ab_variant_trace.sh with
// -DDVP_ABL_FIT=4), by the SAME positive divisor every time.  x -> fl(x / d) is monotone for d > 0 (correct rounding keeps
// order), so { x >= 0 : fl(x / d) < t } is an initial segment of the floats: the test is x <= X* with X* = its largest element
// — found once per pixel from the guess t * d by stepping to the neighbouring floats, the division itself deciding.  Same
// decisions, bit for bit (NaN fails both forms).  Returns -1 when no x >= 0 passes.
DVP_HD float largest_ratio_below(float d, float t) {
	if (!(t > 0.0f)) return -1.0f;
	const float g = t * d;
	uint32_t b = (g >= 0.0f && g <= FLT_MAX) ? f32_bits(g) : 0x7F7FFFFFu;   // start at t * d (FLT_MAX if that overflows)
	while (b > 0u && !(__builtin_bit_cast(float, b) / d < t)) --b;
	while (b < 0x7F7FFFFFu && (__builtin_bit_cast(float, b + 1u) / d < t)) ++b;
	return __builtin_bit_cast(float, b);   // (0 / d = 0 < t: the set is never empty here)
}

// plane through three camera-frame points, as the candidate step builds it (APD.cu:3609-3622)
DVP_HD f4 gn_plane(const f3 A, const f3 B, const f3 C) {
	const f3 AC = mk3(A.x - C.x, A.y - C.y, A.z - C.z);
	const f3 BC = mk3(B.x - C.x, B.y - C.y, B.z - C.z);
	f4 cv;
	cv.x = AC.y * BC.z - BC.y * AC.z;
	cv.y = -(AC.x * BC.z - BC.x * AC.z);
	cv.z = AC.x * BC.y - BC.x * AC.y;
	cv.w = 0.0f;
	return cv;
}
DVP_HD int gn_pair_index(int a, int b) { const int hi = a > b ? a : b, lo = a > b ? b : a; return hi * (hi - 1) / 2 + lo; }

// the plane of candidate draw t again (same operations on the same operands as the candidate step)
DVP_HD f4 cand_plane_of(const FitShared& sh, int t) {
	const uint32_t tr = sh.trip[t];
	const f3 A = sh.sp3[tr & 255u];
	f4 cv = gn_plane(A, sh.sp3[(tr >> 8) & 255u], sh.sp3[(tr >> 16) & 255u]);
	normalize3(&cv);
	cv.w = -(cv.x * A.x + cv.y * A.y + cv.z * A.z);
	return cv;
}

DVP_HD void gen_neighbours_fit_wave(const Dev& d, int px, int py, FitShared& sh) {
	const int W = d.width;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const int wi = d.neighbours_map[center];
	// ---- the list: the directional slots from the search kernel, then the label extension -------------------------
	DVP_LANES(l) { if (l < kGnDirSlots) sh.raw[l] = d.gn_points[(size_t)wi * kGnDirSlots + l]; }
	wave_sync();
	int n_extended = 0;
#if defined(DVP_ABL_FIT) && DVP_ABL_FIT == 1   // timing ablation (wrong results): no label extension
	const int listed = kGnDirSlots;
#else
	const int listed = 1 + gen_neighbours_extend_wave(d, px, py, wi, sh.raw, sh.spv, sh.req_from, sh.req_to, &n_extended);
#endif
	if (d.gn_count[wi] + n_extended <= 3) {   // fewer than four candidates (APD.cu:3562): not reliable, no plane
		if (DVP_LANE0) d.weak_reliable[center] = 0;
		return;
	}
	const DvpCamera cam = load_camera(d, 0);
	const float depth_diff = P.depth_max - P.depth_min;
	const bool by_limit = depth_diff > 0.0f;   // (else: the division, whatever it yields)
	const float inlier_limit = by_limit ? largest_ratio_below(depth_diff, P.ransac_threshold) : 0.0f;
	s2* neighbours = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	const uint32_t site = rng_site(PH_NEIGHBOURS, 0, SUB_RANSAC);
	bool edge_limit = false;   // same draw as in the first half (APD.cu:3366-3374)
	if (P.use_limit) {
		edge_limit = true;
		if (P.use_edge) {
			Rng r_limit(d.seed, (uint32_t)center, rng_site(PH_NEIGHBOURS, 0, SUB_LIMIT));
			if (r_limit.uniform() - FLT_EPSILON < d.complex_[wi]) edge_limit = false;
		}
	}

	// ---- the point tables -------------------------------------------------------------------------------
	if (DVP_LANE0) sh.n_valid = 0;
	wave_sync();
	for (int i0 = 0; i0 < listed; i0 += 64) {
		DVP_LANES(l) {
			s2 r = mks2(-1, -1);
			if (i0 + l < listed) r = sh.raw[i0 + l];
			const bool valid = r.x != -1;
			const int slot = wave_ordered_slot(valid, &sh.n_valid);   // the list without its holes, order kept
			if (valid) sh.slot_of[slot] = (uint8_t)(i0 + l);
		}
	}
	wave_sync();
	const int valid_count = sh.n_valid;
	for (int j0 = 0; j0 < valid_count; j0 += 64) {
		DVP_LANES(l) {
			const int j = j0 + l;
			if (j >= valid_count) continue;
			const s2 sp = sh.raw[sh.slot_of[j]];
			const f4 pl = d.planes[sp.x + sp.y * W];
			float X[3];
			get_3d_point(cam, sp.x, sp.y, pl.w, X);
			sh.spv[j] = sp;
			sh.sp3[j] = mk3(X[0], X[1], X[2]);
			sh.fxy[j] = mk2((sp.x - cam.K[2]) / cam.K[0], (sp.y - cam.K[5]) / cam.K[4]);
		}
	}
	float Xc[3];
	get_3d_point(cam, px, py, d.planes[center].w, Xc);
	const float center_z = Xc[2];
	DVP_LANES(l) { for (int i = l; i < kGnPairWords; i += 64) { sh.seen[i] = 0u; sh.hit[i] = 0u; } sh.mark[l] = -1; }
	wave_sync();

	// ---- 1. the draws; the ones that pass the index and triangle tests are queued ----------------------------
	if (DVP_LANE0) { sh.n_pass = 0; sh.n_walk = 0; }
	wave_sync();
	for (int t0 = 0; t0 < kGnDraws; t0 += 64) {
		DVP_LANES(l) {
			const int t = t0 + l;
			bool pass = false;
			if (t < kGnDraws) {
				const int ai = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t) % (uint32_t)valid_count);
				const int bi = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t + 1u) % (uint32_t)valid_count);
				const int ci = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t + 2u) % (uint32_t)valid_count);
				pass = !(ai == bi || bi == ci || ai == ci) && point_in_triangle(sh.spv[ai], sh.spv[bi], sh.spv[ci], px, py);
				sh.trip[t] = (uint32_t)ai | ((uint32_t)bi << 8) | ((uint32_t)ci << 16) | (pass ? 1u << 24 : 0u);
				sh.cand_info[t] = 0;
			}
			const int slot = wave_ordered_slot(pass, &sh.n_pass);   // plist in draw order
			if (pass) sh.plist[slot] = (uint8_t)t;
		}
	}
	wave_sync();
	const int n_pass = sh.n_pass;
	// ---- 2. + 3. line tests: who asks first, then the walks -----------------------------------------------------
	// Requests in the reference's order: draw t asks (a,b), (b,c), (c,a).  64 requests at a time: a request is
	// a first asker when its pair is neither marked from an earlier batch nor asked by a lower lane of this
	// batch; first askers mark the pair and queue the walk in THEIR orientation.
#if defined(DVP_ABL_FIT) && DVP_ABL_FIT == 2   // timing ablation: no line tests at all (requests + walks)
	if (false) {
#else
	if (edge_limit) {
#endif
		for (int r0 = 0; r0 < 3 * n_pass; r0 += 64) {   // only the draws that passed ask; plist keeps their order
			DVP_LANES(l) {
				const int r = r0 + l, j = r / 3, e = r - 3 * j;
				int idx = -1;
				if (r < 3 * n_pass) {
					const uint32_t tr = sh.trip[sh.plist[j]];
					const int p[4] = { (int)(tr & 255u), (int)((tr >> 8) & 255u), (int)((tr >> 16) & 255u), (int)(tr & 255u) };
					idx = gn_pair_index(p[e], p[e + 1]);
					sh.req_from[l] = (uint8_t)p[e];
					sh.req_to[l] = (uint8_t)p[e + 1];
				}
				sh.req_idx[l] = idx;
			}
			wave_sync();
			DVP_LANES(l) {
				const int idx = sh.req_idx[l];
				if (idx < 0 || ((sh.seen[idx >> 5] >> (idx & 31)) & 1u)) continue;
				bool dup = false;
				for (int m = 0; m < l; ++m) dup = dup || sh.req_idx[m] == idx;
				if (dup) continue;
				sh.walk[wave_counter_add(&sh.n_walk)] = (uint16_t)(sh.req_from[l] | (sh.req_to[l] << 8));
				sh.mark[l] = idx;
			}
			wave_sync();
			DVP_LANES(l) {   // marks after every lane of the batch has looked at the old state
				const int idx = sh.mark[l];
				if (idx >= 0) { wave_bits_or(&sh.seen[idx >> 5], 1u << (idx & 31)); sh.mark[l] = -1; }
			}
			wave_sync();
		}
#if defined(DVP_ABL_FIT) && DVP_ABL_FIT == 3   // timing ablation: requests, but no walks
		const int n_walk = 0;
#else
		const int n_walk = sh.n_walk;
#endif
		for (int j0 = 0; j0 < n_walk; j0 += 64) {
			DVP_LANES(l) {
				const int j = j0 + l;
				if (j >= n_walk) continue;
				const int a = sh.walk[j] & 255, b = sh.walk[j] >> 8;
				if (bresenham_hits_edge(d, sh.spv[a].x, sh.spv[a].y, sh.spv[b].x, sh.spv[b].y)) {
					const int idx = gn_pair_index(a, b);
					wave_bits_or(&sh.hit[idx >> 5], 1u << (idx & 31));
				}
			}
		}
		wave_sync();
	}
	// ---- 4. candidates: one queued draw per lane ------------------------------------------------------------------
	const bool label_test = P.use_label && d.label[center] > 0;
	const float fxc = (px - cam.K[2]) / cam.K[0], fyc = (py - cam.K[5]) / cam.K[4];
	for (int j0 = 0; j0 < n_pass; j0 += 64) {
		DVP_LANES(l) {
			if (j0 + l >= n_pass) continue;
			const int t = sh.plist[j0 + l];
			const uint32_t tr = sh.trip[t];
			const int ai = (int)(tr & 255u), bi = (int)((tr >> 8) & 255u), ci = (int)((tr >> 16) & 255u);
			if (edge_limit) {
				const int i0 = gn_pair_index(ai, bi), i1 = gn_pair_index(bi, ci), i2 = gn_pair_index(ci, ai);
				if (((sh.hit[i0 >> 5] >> (i0 & 31)) | (sh.hit[i1 >> 5] >> (i1 & 31)) | (sh.hit[i2 >> 5] >> (i2 & 31))) & 1u) continue;
			}
			// camera-frame normal of point a — the reference uses a_index for all three normals (APD.cu:3605-3607);
			// fetched again from the plane map instead of a 2 KB table (a third workgroup per CU fits)
			const f4 AN = normal_world_to_cam(cam, d.planes[sh.spv[ai].x + sh.spv[ai].y * W]);
			if (AN.x * AN.x + AN.y * AN.y + AN.z * AN.z < 0.9f) continue;
			const f3 A = sh.sp3[ai];
			f4 cv = gn_plane(A, sh.sp3[bi], sh.sp3[ci]);
			if ((cv.x == 0 && cv.y == 0 && cv.z == 0) || cv.x != cv.x || cv.y != cv.y || cv.z != cv.z) continue;
			normalize3(&cv);
			cv.w = -(cv.x * A.x + cv.y * A.y + cv.z * A.z);
			const bool strong = !(label_test && fabsf(AN.x * cv.x + AN.y * cv.y + AN.z * cv.z) < 0.9f);
			int count = 0;
#if defined(DVP_ABL_FIT) && DVP_ABL_FIT == 4   // timing ablation: no inlier loop
			for (int si = 0; si < 0; ++si) {
#else
			if (by_limit) {   // (two loops: as one loop with the choice inside, both forms are computed and selected)
				for (int si = 0; si < valid_count; ++si) {
					const f2 f = sh.fxy[si];
					const float fit_depth = -cv.w / (cv.x * f.x + cv.y * f.y + cv.z);
					if (fabsf(fit_depth - sh.sp3[si].z) <= inlier_limit) count++;
				}
			} else
			for (int si = 0; si < valid_count; ++si) {
#endif
				const f2 f = sh.fxy[si];
				const float fit_depth = -cv.w / (cv.x * f.x + cv.y * f.y + cv.z);
				if (fabsf(fit_depth - sh.sp3[si].z) / depth_diff < P.ransac_threshold) count++;
			}
			const float fit_depth = -cv.w / (cv.x * fxc + cv.y * fyc + cv.z);
			sh.cand_dist[t] = fabsf(fit_depth - center_z);
			sh.cand_info[t] = 1 | (strong ? 2 : 0) | (count << 8);
		}
	}
	wave_sync();
	// ---- 5. the running best, in draw order (APD.cu:3646-3668) ------------------------------------------------------
	// The reference's update rule — a candidate with at least 6 inliers replaces the best one when it has
	// more inliers, or as many and a strictly smaller distance, and the FIRST "strong" candidate replaces
	// whatever came before, after which only strong ones are looked at — selects, among the strong
	// candidates if there is one (else among all), the one with (most inliers, then smallest distance, then
	// earliest draw).  Every lane folds the draws t = l, l + 64, ... that way, then the 64 partial winners are
	// folded.  A NaN distance does not obey that order: then the scan is done literally.
	f4 best_plane = mk4(0, 0, 0, 0);
	bool has_valid_plane = false;
	{
		DVP_LANES(l) {
			int bt = -1, bstrong = 0, bcount = 0;
			float bdist = 0.0f;
			bool nan = false;
			for (int t = l; t < kGnDraws; t += 64) {
				const int info = sh.cand_info[t];
				if (!(info & 1) || (info >> 8) < 6) continue;
				const int strong = (info >> 1) & 1, count = info >> 8;
				const float dist = sh.cand_dist[t];
				nan = nan || dist != dist;
				if (bt < 0 || strong > bstrong || (strong == bstrong && (count > bcount || (count == bcount && dist < bdist)))) { bt = t; bstrong = strong; bcount = count; bdist = dist; }
			}
			sh.part_t[l] = bt;
			sh.part_nan[l] = nan ? 1 : 0;
		}
		wave_sync();
		bool nan = false;
		int bt = -1, bstrong = 0, bcount = 0;
		float bdist = 0.0f;
		for (int m = 0; m < 64; ++m) {
			nan = nan || sh.part_nan[m] != 0;
			const int t = sh.part_t[m];
			if (t < 0) continue;
			const int info = sh.cand_info[t];
			const int strong = (info >> 1) & 1, count = info >> 8;
			const float dist = sh.cand_dist[t];
			const bool better = bt < 0 || strong > bstrong || (strong == bstrong && (count > bcount || (count == bcount && (dist < bdist || (dist == bdist && t < bt)))));
			if (better) { bt = t; bstrong = strong; bcount = count; bdist = dist; }
		}
		if (!nan) {
			if (bt >= 0) { best_plane = cand_plane_of(sh, bt); has_valid_plane = true; }
		} else {
			bool has_strong_plane = false;
			float min_cost = FLT_MAX;
			int max_count = 3;
			for (int t = 0; t < kGnDraws; ++t) {
				const int info = sh.cand_info[t];
				if (!(info & 1)) continue;
				const bool strong = (info & 2) != 0;
				if (has_strong_plane && !strong) continue;
				const int count = info >> 8;
				if (count < 6) continue;
				const float center_distance = sh.cand_dist[t];
				if (count > max_count || (!has_strong_plane && strong)) {
					if (!has_strong_plane && strong) has_strong_plane = true;
					best_plane = cand_plane_of(sh, t);
					max_count = count;
					min_cost = center_distance;
					has_valid_plane = true;
				} else if (count == max_count) {
					if (center_distance < min_cost) { best_plane = cand_plane_of(sh, t); max_count = count; min_cost = center_distance; }
				}
			}
		}
	}
	if (!has_valid_plane) { if (DVP_LANE0) d.weak_reliable[center] = 0; return; }
	// ---- 6. residuals, stable rank, neighbours ---------------------------------------------------------------------
	for (int j0 = 0; j0 < valid_count; j0 += 64) {
		DVP_LANES(l) {
			const int j = j0 + l;
			if (j >= valid_count) continue;
			const f2 f = sh.fxy[j];
			const float fit_depth = -best_plane.w / (best_plane.x * f.x + best_plane.y * f.y + best_plane.z);
			const float dist = fabsf(fit_depth - sh.sp3[j].z);
			sh.weight[j] = (by_limit ? dist > inlier_limit : dist / depth_diff >= P.ransac_threshold) ? FLT_MAX : dist;
		}
	}
	wave_sync();
	bool any_nan = false;   // a NaN residual (0/0 in the plane equation) has no rank: take the reference's insertion sort literally
	for (int i = 0; i < valid_count; ++i) any_nan = any_nan || sh.weight[i] != sh.weight[i];
	if (any_nan) {
		DVP_LANES(l) {
			if (l != 0) continue;
			for (int i = 0; i < valid_count; ++i)
				if (sh.weight[i] == FLT_MAX) sh.spv[i] = mks2(-1, -1);
			for (int i = 1; i < valid_count; i++) {   // sort_small_weighted (APD.cu:125-138)
				const s2 tp = sh.spv[i];
				const float tw = sh.weight[i];
				int j = i;
				for (; j >= 1 && tw < sh.weight[j - 1]; j--) { sh.spv[j] = sh.spv[j - 1]; sh.weight[j] = sh.weight[j - 1]; }
				sh.spv[j] = tp;
				sh.weight[j] = tw;
			}
			for (int i = 1; i < DVP_NEIGHBOUR_NUM; ++i) neighbours[i] = (i - 1 < valid_count) ? sh.spv[i - 1] : mks2(-1, -1);
			d.weak_reliable[center] = 1;
		}
		return;
	}
	for (int j0 = 0; j0 < valid_count; j0 += 64) {
		DVP_LANES(l) {
			const int j = j0 + l;
			if (j >= valid_count) continue;
			const float w = sh.weight[j];
			int rank = 0;
			for (int i = 0; i < valid_count; ++i) {
				const float wi_ = sh.weight[i];
				rank += (wi_ < w || (wi_ == w && i < j)) ? 1 : 0;
			}
			if (rank < DVP_NEIGHBOUR_NUM - 1) neighbours[1 + rank] = (w == FLT_MAX) ? mks2(-1, -1) : sh.spv[j];
		}
	}
	if (DVP_LANE0) d.weak_reliable[center] = 1;
}

// ---- RANSACToGetFitPlane (APD.cu:4195-4404) with one WAVE per WEAK pixel (round 6; DVP_RANSAC_WAVE=1, not the default: measured no
// faster than the lane kernel — 16.1 vs 15.9 ms per cfg3 pass, 55.6 vs 49.0 at 25 % WEAK; profiles/r06_ab_notes.txt) -------------
// ransac_fit_plane_px (dvp_weak.hpp) is the definition: one lane walks its pixel's 50 draws.  Every draw consumes exactly three
// numbers of the counter-based generator, so draw t is a pure function of t: here lane t owns draw t (lane utilisation of the
// lane-per-pixel launch: 0.18 — pixels differ in how many draws survive the cheap tests, and a survivor costs three line walks
// and a residual loop).  What is ordered in the reference is (i) whose orientation a cached line test takes — the first draw in
// draw order that asks for the pair, all three pairs of a draw before the next draw's (APD.cu:4283-4299) — resolved as in
// gen_neighbours_fit_wave: the requests of the surviving draws in draw order, 64 at a time, a request is a first asker when its
// pair is neither marked by an earlier batch nor asked by a lower lane; (ii) the strict minimum over the draws, taken in draw order.
struct RansacShared {
	s2 sp[DVP_NEIGHBOUR_NUM];
	f3 sp3[DVP_NEIGHBOUR_NUM], spn[DVP_NEIGHBOUR_NUM];
	uint32_t trip[64];             // draw t: a | b << 8 | c << 16
	uint8_t plist[64];             // the draws that passed the cheap tests, in draw order
	int req_idx[64], mark[64];
	uint8_t req_from[64], req_to[64];
	uint16_t walk[64];             // first askers: from | to << 8 (at most 55 unordered pairs)
	uint32_t seen[2], hit[2];      // 55 pair bits
	float cost[64];                // per surviving draw: residual sum, < 0: no plane
	f4 plane[64];
	int n_pass, n_walk, cnt;
};

DVP_HD void ransac_fit_plane_wave(const Dev& d, int px, int py, int iter, RansacShared& sh) {
	const int W = d.width;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	if (d.weak_info[center] != DVP_WEAK) { if (DVP_LANE0) d.fit_planes[center] = d.planes[center]; return; }
	const DvpCamera cam = load_camera(d, 0);
	Rng r_limit(d.seed, (uint32_t)center, rng_site(PH_RANSAC, iter, SUB_LIMIT));
	const uint32_t site = rng_site(PH_RANSAC, iter, SUB_RANSAC);
	bool edge_limit = false;
	if (P.use_limit) {
		edge_limit = true;
		if (P.use_edge) {
			const float complex_val = d.complex_[d.neighbours_map[center]];
			const float rp = r_limit.uniform() - FLT_EPSILON;
			if (rp < complex_val) edge_limit = false;
		}
	}
	if (DVP_LANE0) { sh.cnt = 0; sh.n_pass = 0; sh.n_walk = 0; sh.seen[0] = sh.seen[1] = 0u; sh.hit[0] = sh.hit[1] = 0u; }
	DVP_LANES(l) sh.mark[l] = -1;
	wave_sync();
	// the anchors that exist, in list order
	const s2* nbs = d.neighbours + (size_t)d.neighbours_map[center] * DVP_NEIGHBOUR_NUM;
	DVP_LANES(l) {
		const int i = l + 1;
		s2 tp = mks2(-1, -1);
		if (i < DVP_NEIGHBOUR_NUM) tp = nbs[i];
		const bool valid = i < DVP_NEIGHBOUR_NUM && !(tp.x == -1 || tp.y == -1);
		const int slot = wave_ordered_slot(valid, &sh.cnt);
		if (valid) {
			sh.sp[slot] = tp;
			const f4 pl = d.planes[tp.x + tp.y * W];
			const float depth = depth_from_plane(cam, pl, tp.x, tp.y);
			float X[3];
			get_3d_point(cam, tp.x, tp.y, depth, X);
			sh.sp3[slot] = mk3(X[0], X[1], X[2]);
			sh.spn[slot] = mk3(pl.x, pl.y, pl.z);
		}
	}
	wave_sync();
	const int cnt = sh.cnt;
	if (cnt < 3) { if (DVP_LANE0) d.fit_planes[center] = d.planes[center]; return; }
	// ---- the 50 draws (APD.cu:4262-4330): indices, normals, pixel inside the triangle ----
	DVP_LANES(t) {
		bool pass = false;
		if (t < 50) {
			const int ai = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t) % (uint32_t)cnt);
			const int bi = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t + 1u) % (uint32_t)cnt);
			const int ci = (int)(rand_u32(d.seed, (uint32_t)center, site, 3u * t + 2u) % (uint32_t)cnt);
			sh.trip[t] = (uint32_t)ai | ((uint32_t)bi << 8) | ((uint32_t)ci << 16);
			if (!(ai == bi || bi == ci || ai == ci)) {
				const f3 AN = sh.spn[ai], BN = sh.spn[bi], CN = sh.spn[ci];
				pass = !(AN.x * BN.x + AN.y * BN.y + AN.z * BN.z < 0.9f || AN.x * CN.x + AN.y * CN.y + AN.z * CN.z < 0.9f ||
				         BN.x * CN.x + BN.y * CN.y + BN.z * CN.z < 0.9f) && point_in_triangle(sh.sp[ai], sh.sp[bi], sh.sp[ci], px, py);
			}
		}
		const int slot = wave_ordered_slot(pass, &sh.n_pass);
		if (pass) sh.plist[slot] = (uint8_t)t;
	}
	wave_sync();
	const int n_pass = sh.n_pass;
	// ---- line tests: who asks first (orientation), then the walks ----
	if (edge_limit) {
		for (int r0 = 0; r0 < 3 * n_pass; r0 += 64) {
			DVP_LANES(l) {
				const int r = r0 + l, j = r / 3, e = r - 3 * j;
				int idx = -1;
				if (r < 3 * n_pass) {
					const uint32_t tr = sh.trip[sh.plist[j]];
					const int p[4] = { (int)(tr & 255u), (int)((tr >> 8) & 255u), (int)((tr >> 16) & 255u), (int)(tr & 255u) };
					idx = gn_pair_index(p[e], p[e + 1]);
					sh.req_from[l] = (uint8_t)p[e];
					sh.req_to[l] = (uint8_t)p[e + 1];
				}
				sh.req_idx[l] = idx;
			}
			wave_sync();
			DVP_LANES(l) {
				const int idx = sh.req_idx[l];
				if (idx < 0 || ((sh.seen[idx >> 5] >> (idx & 31)) & 1u)) continue;
				bool dup = false;
				for (int m = 0; m < l; ++m) dup = dup || sh.req_idx[m] == idx;
				if (dup) continue;
				sh.walk[wave_counter_add(&sh.n_walk)] = (uint16_t)(sh.req_from[l] | (sh.req_to[l] << 8));
				sh.mark[l] = idx;
			}
			wave_sync();
			DVP_LANES(l) {
				const int idx = sh.mark[l];
				if (idx >= 0) { wave_bits_or(&sh.seen[idx >> 5], 1u << (idx & 31)); sh.mark[l] = -1; }
			}
			wave_sync();
		}
		const int n_walk = sh.n_walk;   // <= 55
		DVP_LANES(l) {
			if (l >= n_walk) continue;
			const int a = sh.walk[l] & 255, b = sh.walk[l] >> 8;
			if (bresenham_hits_edge(d, sh.sp[a].x, sh.sp[a].y, sh.sp[b].x, sh.sp[b].y)) {
				const int idx = gn_pair_index(a, b);
				wave_bits_or(&sh.hit[idx >> 5], 1u << (idx & 31));
			}
		}
		wave_sync();
	}
	// ---- the surviving draws: plane through the three points, residual over the others ----
	DVP_LANES(l) {
		if (l >= n_pass) continue;
		sh.cost[l] = -1.0f;
		const uint32_t tr = sh.trip[sh.plist[l]];
		const int ai = (int)(tr & 255u), bi = (int)((tr >> 8) & 255u), ci = (int)((tr >> 16) & 255u);
		if (edge_limit) {
			const int i0 = gn_pair_index(ai, bi), i1 = gn_pair_index(bi, ci), i2 = gn_pair_index(ci, ai);
			if (((sh.hit[i0 >> 5] >> (i0 & 31)) | (sh.hit[i1 >> 5] >> (i1 & 31)) | (sh.hit[i2 >> 5] >> (i2 & 31))) & 1u) continue;
		}
		const f3 A = sh.sp3[ai], B = sh.sp3[bi], C = sh.sp3[ci];
		const f3 AC = mk3(A.x - C.x, A.y - C.y, A.z - C.z);
		const f3 BC = mk3(B.x - C.x, B.y - C.y, B.z - C.z);
		f4 cv;
		cv.x = AC.y * BC.z - BC.y * AC.z;
		cv.y = -(AC.x * BC.z - BC.x * AC.z);
		cv.z = AC.x * BC.y - BC.x * AC.y;
		cv.w = 0.0f;
		if ((cv.x == 0 && cv.y == 0 && cv.z == 0) || cv.x != cv.x || cv.y != cv.y || cv.z != cv.z) continue;
		normalize3(&cv);
		cv.w = -(cv.x * A.x + cv.y * A.y + cv.z * A.z);
		float temp_cost = 0.0f;
		for (int si = 0; si < cnt; ++si) {
			if (si == ai || si == bi || si == ci) continue;
			const float fx = (sh.sp[si].x - cam.K[2]) / cam.K[0];
			const float fy = (sh.sp[si].y - cam.K[5]) / cam.K[4];
			const float fit_depth = -cv.w / (cv.x * fx + cv.y * fy + cv.z);
			temp_cost += fabsf(fit_depth - sh.sp3[si].z);
		}
		sh.plane[l] = cv;
		// (a residual sum is >= 0 or NaN; NaN never wins the strict comparison below: stored as "no plane")
		sh.cost[l] = temp_cost == temp_cost ? temp_cost : -1.0f;
	}
	wave_sync();
	// ---- the strict minimum in draw order (APD.cu:4340-4350) ----
	float min_cost = FLT_MAX;
	int best = -1;
	for (int j = 0; j < n_pass; ++j) {
		const float cst = sh.cost[j];
		if (cst >= 0.0f && cst < min_cost) { min_cost = cst; best = j; }
	}
	if (best < 0) {
		if (DVP_LANE0) {
			d.fit_planes[center] = mk4(0, 0, 0, 0);
			if (P.use_radius) d.radius[center] = P.strong_radius;
		}
		return;
	}
	f4 best_plane = sh.plane[best];
	const uint32_t btr = sh.trip[sh.plist[best]];
	const int use_a = (int)(btr & 255u), use_b = (int)((btr >> 8) & 255u), use_c = (int)((btr >> 16) & 255u);
	const float depth = depth_from_plane(cam, d.planes[center], px, py);
	const f4 vdir = view_direction(cam, px, py, depth);
	const float dp = best_plane.x * vdir.x + best_plane.y * vdir.y + best_plane.z * vdir.z;
	if (dp > 0) { best_plane.x = -best_plane.x; best_plane.y = -best_plane.y; best_plane.z = -best_plane.z; best_plane.w = -best_plane.w; }
	if (DVP_LANE0) d.fit_planes[center] = best_plane;
	if (P.use_radius) {
		const s2 A = sh.sp[use_a], B = sh.sp[use_b], C = sh.sp[use_c];
		const int radius = ransac_patch_radius(d, px, py, center, A, B, C, edge_limit);
		if (DVP_LANE0) d.radius[center] = radius < P.strong_radius ? 0 : radius;
	}
}

}  // namespace dvp
#endif
