// dvp_engine.hip — gfx950 kernels + the C ABI of include/dvp_mvs.h.
//
// Replaces APD::CudaSpaceInitialization / SetDataPassHelperInCuda / RunPatchMatch
// (/root/reference/APD.cpp:1497-1613, 1670-1704; APD.cu:4406-4532).  One context owns one HIP
// stream and every device buffer of one reference view; launches are queued back to back on that
// stream (the reference calls cudaDeviceSynchronize after each of its 16+5*iters launches).
#include "dvp_stages.hpp"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <chrono>
#include <algorithm>
#include <cmath>

using namespace dvp;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct LaunchArgs {
	int tiles_x, tiles, rows, half, colour, iter;
};

template <int STAGE, int SMP, int MV = 32>
__device__ __forceinline__ void stage_body(const Dev& d, const LaunchArgs& a) {
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	int px, py;
	unsigned long long n = 0;
	// per-lane (w, w*ref) table of the hoisted patch context: [tap][lane] in LDS (72 KiB / workgroup)
	__shared__ f2 lds_tab[stage_uses_tab(STAGE) ? kTaps * kTaps * 256 : 1];
	const PatchTab tab{&lds_tab[stage_uses_tab(STAGE) ? threadIdx.x : 0], 256};
	if (block_to_pixel(blockIdx.x, lane, wave, a.tiles_x, a.tiles, a.rows, a.half, a.colour, d.width, d.height, &px, &py))
		run_pixel<STAGE, SMP, MV>(d, px, py, a.iter, d.eval_counter ? &n : nullptr, tab);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}

// Compacted launch for the weak-pixel path: lane t owns the t-th WEAK pixel of the list segment.
// WEAK pixels are 1-30 % of a view; a lane-per-image-pixel launch leaves 70-99 % of the lanes idle.
struct ListArgs { int base, count, iter, covered_rows, group, run; };   // group: pixels per wave of the weak update's evaluation launches
// XCD-aware block -> list-block map.  Workgroup b runs on XCD b % 8 (observed placement, a speed matter
// only); the list is in super-tile order, so giving every XCD RUNS of kListRun consecutive list blocks
// (instead of every 8th block) keeps the workgroups that share anchors — and the source-image lines
// their anchor sub-patches gather — behind one L2.  Bijection on [0, nblocks): whole groups of
// 8 * kListRun blocks are permuted, the ragged tail keeps its order.
#ifndef DVP_LIST_RUN
#define DVP_LIST_RUN 16
#endif
constexpr int kListRun = DVP_LIST_RUN;
// The wave-per-pixel kernels (workgroup = 4 WEAK pixels) use shorter runs: 1 .. 64 blocks measure the same (weak update 400-402 ms
// per cfg3 pass), 256 blocks 405, the 1 024 of round 2 (the lane kernels' run in pixels) 419 — a run longer than what an XCD has
// in flight only delays the neighbours.
#ifndef DVP_WAVE_RUN
#define DVP_WAVE_RUN 16
#endif
constexpr int kWaveRun = DVP_WAVE_RUN;
__device__ __forceinline__ int list_block(int b, int nblocks, int run) {
	const int group = 8 * run;
	if (b >= nblocks / group * group) return b;
	const int g = b / group, r = b - g * group;
	const int xcd = r & 7, i = r >> 3;
	return g * group + xcd * run + i;
}
template <int STAGE, int SMP, int MV = 32>
__device__ __forceinline__ void stage_body_list(const Dev& d, const ListArgs& a) {
	__shared__ f2 lds_tab[stage_uses_tab(STAGE) ? kTaps * kTaps * 256 : 1];
	const PatchTab tab{&lds_tab[stage_uses_tab(STAGE) ? threadIdx.x : 0], 256};
	const int t = list_block(blockIdx.x, gridDim.x, kListRun) * 256 + threadIdx.x;
	unsigned long long n = 0;
	if (t < a.count) {
		const int center = d.weak_list[a.base + t];
		const int py = center / d.width, px = center - py * d.width;
		// red/black launches never reach rows beyond the reference's half grid (APD.cu:4421-4424)
		if (!stage_is_half_c(STAGE) || py < a.covered_rows)
			run_pixel<STAGE, SMP, MV>(d, px, py, a.iter, d.eval_counter ? &n : nullptr, tab);
	}
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
// The same launch sites with ONE wave per workgroup (18 KB of LDS each): tile t = the workgroups b with
// (b >> 5) * 8 + (b & 7) == t, its row (b >> 3) & 3 — the four rows of a tile stay on the XCD block_to_pixel's map gives the tile.
// No wave waits for the slowest of its workgroup before the LDS is given back: the strong update's evaluation and refinement
// kernels gain 11 ms per cfg3 pass (DepthToWeak nothing: stays on 256-lane workgroups).
template <int STAGE, int SMP, int MV = 32>
__device__ __forceinline__ void stage_body64(const Dev& d, const LaunchArgs& a) {
	const int lane = threadIdx.x;
	const int b = blockIdx.x;
	const int tile = (b >> 5) * 8 + (b & 7), wave = (b >> 3) & 3;
	int px, py;
	unsigned long long n = 0;
	__shared__ f2 lds_tab[kTaps * kTaps * 64];
	const PatchTab tab{&lds_tab[lane], 64};
	if (block_to_pixel(tile, lane, wave, a.tiles_x, a.tiles, a.rows, a.half, a.colour, d.width, d.height, &px, &py))
		run_pixel<STAGE, SMP, MV>(d, px, py, a.iter, d.eval_counter ? &n : nullptr, tab);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
#define DVP_KERNEL64(NAME, STAGE, MINW)                                                                    \
	extern "C" __global__ void __launch_bounds__(64, MINW) NAME(const Dev d, const LaunchArgs a) {          \
		stage_body64<STAGE, 0>(d, a);                                                                       \
	}                                                                                                       \
	extern "C" __global__ void __launch_bounds__(64, MINW) NAME##_exact(const Dev d, const LaunchArgs a) {  \
		stage_body64<STAGE, 1>(d, a);                                                                       \
	}
#define DVP_KERNEL_LIST_MV(NAME, STAGE, MINW, MV)                                                         \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME(const Dev d, const ListArgs a) {         \
		stage_body_list<STAGE, 0, MV>(d, a);                                                               \
	}                                                                                                      \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME##_exact(const Dev d, const ListArgs a) { \
		stage_body_list<STAGE, 1, MV>(d, a);                                                               \
	}
#define DVP_KERNEL_LIST(NAME, STAGE, MINW)                                                                \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME(const Dev d, const ListArgs a) {         \
		stage_body_list<STAGE, 0>(d, a);                                                                   \
	}                                                                                                      \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME##_exact(const Dev d, const ListArgs a) { \
		stage_body_list<STAGE, 1>(d, a);                                                                   \
	}

// One named kernel per launch site so that rocprofv3 --kernel-trace shows the reference's names.
// NAME: sampler 0 (8-bit interpolation weights, default); NAME_exact: sampler 1.
#define DVP_KERNEL(NAME, STAGE, MINW)                                                                      \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME(const Dev d, const LaunchArgs a) {         \
		stage_body<STAGE, 0>(d, a);                                                                         \
	}                                                                                                       \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME##_exact(const Dev d, const LaunchArgs a) { \
		stage_body<STAGE, 1>(d, a);                                                                         \
	}

// same launch site, private per-view arrays sized for MV views (strong update with S <= 8)
#define DVP_KERNEL_MV(NAME, STAGE, MINW, MV)                                                               \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME(const Dev d, const LaunchArgs a) {         \
		stage_body<STAGE, 0, MV>(d, a);                                                                     \
	}                                                                                                       \
	extern "C" __global__ void __launch_bounds__(256, MINW) NAME##_exact(const Dev d, const LaunchArgs a) { \
		stage_body<STAGE, 1, MV>(d, a);                                                                     \
	}

DVP_KERNEL(dvp_gen_edge_inform, DVP_ST_GEN_EDGE_INFORM, 1)
#ifndef DVP_LB_HEAVY
// min waves/SIMD the NCC kernels are compiled for: 2 = 256 VGPRs per lane (the pipelined 36-tap
// evaluation keeps 12 sixteen-byte gathers + the next 12 footprints live) and two 72 KiB patch
// tables per CU in LDS.
#define DVP_LB_HEAVY 2
#endif
DVP_KERNEL(dvp_random_init, DVP_ST_RANDOM_INIT, DVP_LB_HEAVY)
DVP_KERNEL(dvp_strong_update, DVP_ST_STRONG_UPDATE, DVP_LB_HEAVY)
DVP_KERNEL_MV(dvp_strong_update_v8, DVP_ST_STRONG_UPDATE, DVP_LB_HEAVY, kNarrowViews)
DVP_KERNEL_MV(dvp_strong_update_v16, DVP_ST_STRONG_UPDATE, DVP_LB_HEAVY, 16)
// split strong update (dvp_strong.hpp): evaluations of the 17 snapshot planes, decisions, refinement (S <= 16)
DVP_KERNEL64(dvp_strong_eval, kStageStrongEval, DVP_LB_HEAVY)
// dvp_strong_eval with the (pixel, slot) items of the wave's 64 pixels compacted over its lanes.  With a pixel per lane a
// wave makes 17 trips x S views whatever its lanes hold: WEAK and off-image lanes idle, edge pixels have 9 slots instead of
// 17, and 11 % of the remaining planes are bitwise repeats of an earlier slot (strong_slot_sources) — 18 % of the lane-trips
// at cfg3.  Here every lane first builds its pixel's patch table (LDS, [tap][pixel]) and lists its distinct slots; the
// items are then numbered across the wave (prefix sum of the counts) and taken 64 per round: lane l evaluates item
// round * 64 + l = (pixel p, slot s) against the S views with p's table column and p's reference sums (fetched from the
// owner lane by ds_bpermute).  Same evaluations as strong_eval_px, hence the same bits.  LDS: 18 KB table + 1088 B of item
// list = 8 workgroups per CU, as before.
template <int SMP>
__device__ __forceinline__ void strong_eval_items_body(const Dev& d, const LaunchArgs& a) {
	const int lane = threadIdx.x;
	const int b = blockIdx.x;
	const int tile = (b >> 5) * 8 + (b & 7), wave = (b >> 3) & 3;
	__shared__ f2 lds_tab[kTaps * kTaps * 64];
	__shared__ uint8_t items[64 * kSlotCount];
	int px = 0, py = 0;
	bool active = block_to_pixel(tile, lane, wave, a.tiles_x, a.tiles, a.rows, a.half, a.colour, d.width, d.height, &px, &py);
	if (active && d.weak_info[px + py * d.width] == DVP_WEAK) active = false;
	PatchCtx c;
	c.tab = PatchTab{&lds_tab[lane], 64};
	c.sum_ref = c.sum_ref_ref = c.wsum = 0.0f;
	c.radius = c.inc = c.fast = 0;
	uint32_t uniq = 0;
	if (active) {
		const int center = px + py * d.width;
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, c.tab, &c);
		uint32_t w[3];
		uniq = strong_slot_sources(d, center, w);
		const size_t Lh = (size_t)d.half_w * (size_t)d.height;
		uint32_t* dup = reinterpret_cast<uint32_t*>(d.strong_rec) + half_index(d, px, py);
		dup[SR_DUP * Lh] = w[0]; dup[(SR_DUP + 1) * Lh] = w[1]; dup[(SR_DUP + 2) * Lh] = w[2];
	}
	// Item order: pixel-major — all distinct slots of pixel 0, then pixel 1, ...; exclusive prefix sum of the lanes' item
	// counts.  (Rank-major — the first distinct slot of every pixel, then the second, ... — keeps neighbouring pixels with the
	// same direction's sample side by side like the pixel-per-lane kernel: L2 hit rate of the launch site 0.58 -> 0.84 and 65
	// -> 48 B of fabric traffic per evaluation, but the same time at cfg3 (467 vs 466 ms) and 3.8 % more at cfg2 (148 vs 143
	// ms): the kernel issues VALU instructions, it does not wait for lines, and the lanes of one pixel share its table column.)
	const int n = __popc(uniq);
	int incl = n;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(incl, o, 64);
		if (lane >= o) incl += t;
	}
	const int off = incl - n;
	const int total = __shfl(incl, 63, 64);
	{
		int k = 0;
		for (uint32_t m = uniq; m; m &= m - 1, ++k) items[off + k] = (uint8_t)lane;
	}
	__syncthreads();
	unsigned long long cnt = 0;
	for (int i0 = 0; i0 < total; i0 += 64) {
		const bool valid = i0 + lane < total;
		const int i = valid ? i0 + lane : total - 1;   // the tail of the last round repeats its last item without storing
		const int pl = items[i];
		const int k = i - __shfl(off, pl, 64);
		uint32_t m = (uint32_t)__shfl((int)uniq, pl, 64);
		for (int j = 0; j < k; ++j) m &= m - 1;
		const int slot = dvp_ctz(m);
		PatchCtx ci;
		ci.tab = PatchTab{&lds_tab[pl], 64};
		ci.sum_ref = __shfl(c.sum_ref, pl, 64);
		ci.sum_ref_ref = __shfl(c.sum_ref_ref, pl, 64);
		ci.wsum = __shfl(c.wsum, pl, 64);
		ci.radius = __shfl(c.radius, pl, 64);
		ci.inc = __shfl(c.inc, pl, 64);
		ci.fast = __shfl(c.fast, pl, 64);
		const int ipx = __shfl(px, pl, 64), ipy = __shfl(py, pl, 64);
		strong_eval_item<SMP>(d, ci, ipx, ipy, slot, valid, d.eval_counter ? &cnt : nullptr);
	}
	if (d.eval_counter && cnt) atomicAdd(d.eval_counter, cnt);
}
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_strong_eval_items(const Dev d, const LaunchArgs a) { strong_eval_items_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_strong_eval_items_exact(const Dev d, const LaunchArgs a) { strong_eval_items_body<1>(d, a); }
DVP_KERNEL64(dvp_strong_refine, kStageStrongRefine, DVP_LB_HEAVY)
DVP_KERNEL64(dvp_strong_refine_lanes, kStageStrongRefineLanes, DVP_LB_HEAVY)
template <int MV>
__device__ __forceinline__ void strong_decide_body(const Dev& d, const LaunchArgs& a) {
	int px, py;
	if (!block_to_pixel(blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, a.tiles_x, a.tiles, a.rows, a.half, a.colour, d.width, d.height, &px, &py)) return;
	if (d.weak_info[px + py * d.width] != DVP_WEAK) strong_decide_px<MV>(d, px, py, a.iter);
}
// the cost vectors of the 8 directions live in registers: one instantiation per view-count bracket
// (4 waves per SIMD: the kernel waits for its cost loads 70 % of the time; 128 VGPRs cost the v10 instantiation 20 spilled registers and
// win 1.7 ms of 10 per cfg3 launch; 5 waves the same, 6 less)
#ifndef DVP_LB_DECIDE
#define DVP_LB_DECIDE 4
#endif
#define DVP_DECIDE_KERNEL(MV) extern "C" __global__ void __launch_bounds__(256, DVP_LB_DECIDE) dvp_strong_decide_v##MV(const Dev d, const LaunchArgs a) { strong_decide_body<MV>(d, a); }
// (Round 6 built the same decisions with the wave's 64 cost records staged in LDS by coalesced loads — 39 KB per wave at S = 9, i.e.
// four waves per CU: 17.8 ms per cfg3 launch against 8.0 for the plain loads below on the pixel-major records; removed,
// profiles/r06_ab_notes.txt.)
DVP_DECIDE_KERNEL(4)
DVP_DECIDE_KERNEL(6)
DVP_DECIDE_KERNEL(8)
DVP_DECIDE_KERNEL(10)
DVP_DECIDE_KERNEL(12)
DVP_DECIDE_KERNEL(16)
DVP_KERNEL(dvp_get_depth_normal, DVP_ST_GET_DEPTH_NORMAL, 1)
DVP_KERNEL(dvp_filter_strong, DVP_ST_FILTER_STRONG, 1)
DVP_KERNEL(dvp_depth_to_weak, DVP_ST_DEPTH_TO_WEAK, DVP_LB_HEAVY)
DVP_KERNEL(dvp_local_refine, DVP_ST_LOCAL_REFINE, DVP_LB_HEAVY)
DVP_KERNEL(dvp_depth_to_weak_refine, kStageSweeps, DVP_LB_HEAVY)   // the two sweeps in one launch (dvp_run_patchmatch)
// ... and the same launch site as view-compacted passes (dvp_strong.hpp: sweep_*)
template <int WHAT>
__device__ __forceinline__ void sweep_light_body(const Dev& d, const LaunchArgs& a) {
	int px, py;
	if (!block_to_pixel(blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py)) return;
	if (WHAT != 0 && d.sweep_row1 > 0 && (py < d.sweep_row0 || py >= d.sweep_row1)) return;   // a band's decision pass
	if (WHAT == 0) sweep_prepare_px(d, px, py);
	else if (WHAT == 1) sweep_decide1_px(d, px, py);
	else sweep_decide2_px(d, px, py);
}
extern "C" __global__ void __launch_bounds__(256) dvp_sweep_prepare(const Dev d, const LaunchArgs a) { sweep_light_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(256) dvp_sweep_decide1(const Dev d, const LaunchArgs a) { sweep_light_body<1>(d, a); }
extern "C" __global__ void __launch_bounds__(256) dvp_sweep_decide2(const Dev d, const LaunchArgs a) { sweep_light_body<2>(d, a); }
// Evaluation pass: workgroup = one wave = one 64 x 4 tile of the launch map, grid.y = source view (dispatched view after view:
// the workgroups in flight gather from ONE image).  The tile's pixels that take part — they selected the view with a weight
// > 0 — are compacted into a list in LDS (ballot ranks, row after row) and taken 64 per round, so that every lane of a round
// evaluates (71 % of the pixels select a given view at cfg3: 2.8 rounds of 64 instead of 4 rows with 29 % of the lanes idle).
// a.iter = stage (0: central window + LocalRefine's extra slot, 1: the rest of the line for pixels with a central peak).
#ifndef DVP_SWEEP_ROWS
#define DVP_SWEEP_ROWS 14
#endif
constexpr int kSweepRows = DVP_SWEEP_ROWS;   // rows of a dvp_sweep_eval tile (64 x kSweepRows pixels; 14: patch table 18 KB + list 1.75 KB + two cameras = 20 KB, 8 workgroups per CU)
template <int SMP>
__device__ __forceinline__ void sweep_eval_body(const Dev& d, const LaunchArgs& a) {
	const int lane = threadIdx.x;
	const int v = blockIdx.y;
	__shared__ f2 lds_tab[kTaps * kTaps * 64];
	__shared__ uint16_t list[64 * kSweepRows];   // tile-local pixel index (row * 64 + x): 18 KB + 2 KB = 8 workgroups per CU
	__shared__ DvpCamera cams[2];                // reference camera, camera of the launch's view
	const PatchTab tab{&lds_tab[lane], 64};
	static_assert(sizeof(DvpCamera) == 112, "28 dwords");
	if (lane < 56) {
		const int which = lane >= 28 ? 1 : 0;
		reinterpret_cast<uint32_t*>(&cams[which])[lane - 28 * which] = reinterpret_cast<const uint32_t*>(d.cameras + (which ? v + 1 : 0))[lane - 28 * which];
	}
	// tile of this workgroup: the strip map of block_to_pixel (XCD b % 8 owns a column of a strip of 8 tile columns) over
	// tiles of kSweepRows rows; a.tiles_x / a.tiles = that grid
	const int tiles_y = a.tiles / a.tiles_x;
	int tx, ty;
	{
		const int per_strip = 8 * tiles_y;
		const int st = (int)blockIdx.x / per_strip, rem = (int)blockIdx.x - st * per_strip;
		const int w_last = a.tiles_x - st * 8;
		if (w_last >= 8) { ty = rem >> 3; tx = st * 8 + (rem & 7); }
		else { ty = rem / w_last; tx = st * 8 + (rem - ty * w_last); }
	}
	ty += a.rows;   // first tile row of the launch (a band of the image; 0: all of it)
	const int x = tx * 64 + lane;
	int n = 0;
	for (int r = 0; r < kSweepRows; ++r) {
		const int y = ty * kSweepRows + r;
		const bool go = x < d.width && y < d.height && sweep_go(d, x + y * d.width, v, a.iter);
		const unsigned long long m = __ballot(go);
		if (go) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(r * 64 + lane);
		n += __popcll(m);
	}
	__syncthreads();
	unsigned long long cnt = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		if (i0 + lane < n) {
			const int e = list[i0 + lane];
			sweep_eval_px<SMP>(d, tx * 64 + (e & 63), ty * kSweepRows + (e >> 6), v, a.iter, tab, d.eval_counter ? &cnt : nullptr, cams);
		}
	}
	if (d.eval_counter && cnt) atomicAdd(d.eval_counter, cnt);
}
// The 6-pixel frame of the view-compacted form (UNKNOWN for DepthToWeak, a full LocalRefine: the fused kernel's per-pixel
// code) over the frame's pixels only: lane t of the launch = the t-th frame pixel (6 rows on top, 6 at the bottom, 2 x 6
// columns between them).  The full-grid launch kept 8 800 waves resident for 6 active lanes each at 6208 x 4128 (6.6 ms).
template <int SMP>
__device__ __forceinline__ void sweep_border_body(const Dev& d, const LaunchArgs& a) {
	__shared__ f2 lds_tab[kTaps * kTaps * 64];
	const PatchTab tab{&lds_tab[threadIdx.x], 64};
	const int W = d.width, H = d.height;
	const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
	int px, py;
	if (t < 12ll * W) {
		const int r = (int)(t / W);
		px = (int)(t - (long long)r * W);
		py = r < 6 ? r : H - 12 + r;
	} else {
		const long long u = t - 12ll * W;
		const int r = (int)(u / 12), cidx = (int)(u - 12ll * r);
		if (r >= H - 12) return;
		py = 6 + r;
		px = cidx < 6 ? cidx : W - 12 + cidx;
	}
	unsigned long long n = 0;
	run_pixel<kStageSweeps, SMP>(d, px, py, kSweepBorderOnly, d.eval_counter ? &n : nullptr, tab);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_sweep_border(const Dev d, const LaunchArgs a) { sweep_border_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_sweep_border_exact(const Dev d, const LaunchArgs a) { sweep_border_body<1>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_sweep_eval(const Dev d, const LaunchArgs a) { sweep_eval_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_HEAVY) dvp_sweep_eval_exact(const Dev d, const LaunchArgs a) { sweep_eval_body<1>(d, a); }

// the weak-path launch sites: one lane per entry of the WEAK-pixel list
DVP_KERNEL_LIST(dvp_find_nearest_strong_list, DVP_ST_FIND_NEAREST_STRONG, 1)
// GenNeighbours' directional search: one lane per WEAK pixel, the list of found points in LDS (one column per lane)
extern "C" __global__ void __launch_bounds__(256) dvp_gen_neighbours_list(const Dev d, const ListArgs a) {
	__shared__ s2 pts[kGnDirSlots * 256];
	const int t = list_block(blockIdx.x, gridDim.x, kListRun) * 256 + threadIdx.x;
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const int py = center / d.width, px = center - py * d.width;
	gen_neighbours_px(d, px, py, pts + threadIdx.x, 256);
}
DVP_KERNEL_LIST(dvp_neighbour_update_list, DVP_ST_NEIGHBOUR_UPDATE, 1)
DVP_KERNEL_LIST(dvp_ransac_fit_plane_list, DVP_ST_RANSAC_FIT, 1)
// ... and the same launch site with one wave per WEAK pixel, lane = draw (dvp_weak_wave.hpp: ransac_fit_plane_wave; DVP_RANSAC_WAVE=1 —
// measured no faster than the lane kernel, whose 0.18 lane utilisation turns out not to be what it waits for: kept as the record of that)
extern "C" __global__ void __launch_bounds__(64) dvp_ransac_fit_plane_wave(const Dev d, const ListArgs a) {
	__shared__ RansacShared sh[1];
	const int t = list_block(blockIdx.x, gridDim.x, kWaveRun * 4);
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const int py = center / d.width, px = center - py * d.width;
	ransac_fit_plane_wave(d, px, py, a.iter, sh[0]);
}

// Black/RedPixelUpdateWeak (APD.cu:4487-4489): one WAVE per WEAK pixel of the list segment, per-pixel state in LDS
// (dvp_weak_wave.hpp)
template <int SMP, int FMT, int TAB>
__device__ __forceinline__ void weak_wave_body(const Dev& d, const ListArgs& a) {
	// one wave = one WEAK pixel = one workgroup (10 KB of LDS): with four pixels per workgroup the LDS of a finished pixel waited
	// for the slowest of the four (weak update 400 -> 376 ms per cfg3 pass); the XCD runs keep their length in pixels
	__shared__ WeakSharedT<TAB> sh[1];
	const int wave = 0;
	const int t = list_block(blockIdx.x, gridDim.x, kWaveRun * 4);
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const int py = center / d.width, px = center - py * d.width;
	if (py >= a.covered_rows) return;   // rows beyond the reference's half grid (APD.cu:4421-4424)
	if (d.weak_info[center] != DVP_WEAK) return;   // NeigbourUpdate turned it UNKNOWN since the list was built (APD.cu:3119-3123)
	unsigned long long n = 0;
	weak_update_wave<SMP, FMT, TAB>(d, px, py, a.iter, d.eval_counter ? &n : nullptr, sh[wave]);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
// first half of GenNeighbours (directional anchor search + label extension): one wave per WEAK pixel, the tries of a
// direction over the lanes (dvp_weak_wave.hpp: gen_neighbours_search_wave)
#ifndef DVP_LB_GN
#define DVP_LB_GN 8   // waves per SIMD (64 VGPRs + 164 B scratch; measured 86 ms per cfg3 pass, 4 waves without scratch: 105 ms)
#endif
extern "C" __global__ void __launch_bounds__(256, DVP_LB_GN) dvp_gen_neighbours_search(const Dev d, const ListArgs a) {
	__shared__ GnShared sh[4];
	const int wave = threadIdx.x >> 6;
	const int t = list_block(blockIdx.x, gridDim.x, kWaveRun) * 4 + wave;
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	if (d.weak_info[center] != DVP_WEAK) return;
	const int py = center / d.width, px = center - py * d.width;
	gen_neighbours_search_wave(d, px, py, sh[wave]);
}
// second half of GenNeighbours (RANSAC plane + ranking): one wave per WEAK pixel, point tables in LDS
// (3 workgroups per CU is what the 12.6 KB of LDS per wave allow; stated, the compiler fits the kernel in 108 VGPRs, left alone it takes 179: 2 waves per SIMD)
extern "C" __global__ void __launch_bounds__(64, 3) dvp_gen_neighbours_fit(const Dev d, const ListArgs a) {
	__shared__ FitShared sh[1];   // (one wave per workgroup, as the weak update: 65 -> 59 ms per cfg3 launch)
	const int wave = 0;
	const int t = list_block(blockIdx.x, gridDim.x, kWaveRun * 4);
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	if (d.weak_info[center] != DVP_WEAK) return;
	const int py = center / d.width, px = center - py * d.width;
	gen_neighbours_fit_wave(d, px, py, sh[wave]);
}

#ifndef DVP_LB_WEAK
#define DVP_LB_WEAK 4   // waves per SIMD the wave kernel is compiled for (LDS: 34 KB per workgroup -> 4 workgroups per CU; 128 VGPRs)
#endif
// (no suffix: the anchor sub-patches' reference side comes from the pass' table, Dev::anchor_tab; _notab: formed per item —
// the definition, used when the table does not fit or DVP_WEAK_ANCHOR_TAB=0)
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave(const Dev d, const ListArgs a) { weak_wave_body<0, 0, 1>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_exact(const Dev d, const ListArgs a) { weak_wave_body<1, 0, 1>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_notab(const Dev d, const ListArgs a) { weak_wave_body<0, 0, 0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_exact_notab(const Dev d, const ListArgs a) { weak_wave_body<1, 0, 0>(d, a); }
// the same launch site reading the byte planes (Dev::images8: all images 8-bit exact)
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_u8(const Dev d, const ListArgs a) { weak_wave_body<0, 1, 1>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_exact_u8(const Dev d, const ListArgs a) { weak_wave_body<1, 1, 1>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_u8_notab(const Dev d, const ListArgs a) { weak_wave_body<0, 1, 0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) dvp_weak_update_wave_exact_u8_notab(const Dev d, const ListArgs a) { weak_wave_body<1, 1, 0>(d, a); }
// The same launch site as EIGHT launches (dvp_weak_phased.hpp): the evaluation launches E0 / E1 / E2a / E2b take one wave per
// GROUP of a.group consecutive WEAK pixels of the list, the per-pixel decisions D1 / D2 / D3 and the final plain-NCC cost E3
// run one LANE per WEAK pixel.
template <int SMP, int FMT, int MODE>
__device__ __forceinline__ void weak_group_body(const Dev& d, const ListArgs& a) {
	constexpr int GRP = MODE == 0 ? kGrpWide : kGrp;
	__shared__ WeakGroupSharedT<GRP> sh;
	const int G = a.group < GRP ? a.group : GRP;
	const int run = a.run / G;          // XCD runs of a.run pixels (the one-wave kernel's: kWaveRun * 4)
	const int blk = list_block(blockIdx.x, gridDim.x, run < 1 ? 1 : run);
	if (blk * G >= a.count) return;
	if (threadIdx.x < (unsigned)GRP) {
		const int t = blk * G + (int)threadIdx.x;
		int center = -1;
		if ((int)threadIdx.x < G && t < a.count) {
			center = d.weak_list[a.base + t];
			// rows beyond the reference's half grid (APD.cu:4421-4424); NeigbourUpdate turned it UNKNOWN since the list was built (APD.cu:3119-3123)
			if (center / d.width >= a.covered_rows || d.weak_info[center] != DVP_WEAK) center = -1;
		}
		sh.center[threadIdx.x] = center;
	}
	wave_sync();
	unsigned long long n = 0;
	weak_group_eval<SMP, FMT, MODE, GRP>(d, G, d.eval_counter ? &n : nullptr, sh);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
#define DVP_WEAK_PHASE_KERNELS(NAME, MODE)                                                                                                              \
	extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) NAME(const Dev d, const ListArgs a) { weak_group_body<0, 0, MODE>(d, a); }           \
	extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) NAME##_exact(const Dev d, const ListArgs a) { weak_group_body<1, 0, MODE>(d, a); }   \
	extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) NAME##_u8(const Dev d, const ListArgs a) { weak_group_body<0, 1, MODE>(d, a); }      \
	extern "C" __global__ void __launch_bounds__(64, DVP_LB_WEAK) NAME##_exact_u8(const Dev d, const ListArgs a) { weak_group_body<1, 1, MODE>(d, a); }
DVP_WEAK_PHASE_KERNELS(dvp_weak_eval_candidates, 0)
DVP_WEAK_PHASE_KERNELS(dvp_weak_eval_planes, 1)
DVP_WEAK_PHASE_KERNELS(dvp_weak_eval_first_view, 2)
DVP_WEAK_PHASE_KERNELS(dvp_weak_eval_survivors, 3)
template <int PART>
__device__ __forceinline__ void weak_phase_lane_body(const Dev& d, const ListArgs& a) {
	const int t = list_block(blockIdx.x, gridDim.x, kListRun) * 256 + threadIdx.x;
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const int py = center / d.width, px = center - py * d.width;
	if (py >= a.covered_rows) return;
	if (d.weak_info[center] != DVP_WEAK) return;
	if (PART == 0) weak_d1_px(d, px, py, a.iter);
	else if (PART == 1) weak_d2_px(d, px, py, a.iter);
	else weak_d3_px(d, px, py);
}
#ifndef DVP_LB_WEAK_DECIDE
#define DVP_LB_WEAK_DECIDE 2   // 256-lane workgroups per SIMD... (waves per SIMD = this; the kernels wait for scattered loads)
#endif
extern "C" __global__ void __launch_bounds__(256, DVP_LB_WEAK_DECIDE) dvp_weak_select_views(const Dev d, const ListArgs a) { weak_phase_lane_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(256, DVP_LB_WEAK_DECIDE) dvp_weak_make_hypotheses(const Dev d, const ListArgs a) { weak_phase_lane_body<1>(d, a); }
extern "C" __global__ void __launch_bounds__(256, DVP_LB_WEAK_DECIDE) dvp_weak_adopt(const Dev d, const ListArgs a) { weak_phase_lane_body<2>(d, a); }
// E3: the per-lane evaluator of the strong path over the WEAK list (one-wave workgroups, 18 KB of patch table each)
template <int SMP>
__device__ __forceinline__ void weak_final_cost_body(const Dev& d, const ListArgs& a) {
	__shared__ f2 lds_tab[kTaps * kTaps * 64];
	const PatchTab tab{&lds_tab[threadIdx.x], 64};
	const int t = list_block(blockIdx.x, gridDim.x, kListRun * 4) * 64 + threadIdx.x;
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const int py = center / d.width, px = center - py * d.width;
	if (py >= a.covered_rows) return;
	if (d.weak_info[center] != DVP_WEAK) return;
	unsigned long long n = 0;
	weak_final_cost_px<SMP>(d, px, py, tab, d.eval_counter ? &n : nullptr);
	if (d.eval_counter && n) atomicAdd(d.eval_counter, n);
}
extern "C" __global__ void __launch_bounds__(64, 2) dvp_weak_final_cost(const Dev d, const ListArgs a) { weak_final_cost_body<0>(d, a); }
extern "C" __global__ void __launch_bounds__(64, 2) dvp_weak_final_cost_exact(const Dev d, const ListArgs a) { weak_final_cost_body<1>(d, a); }

// the pass' table of anchor reference sides: thread = (WEAK list entry, view, anchor), anchor fastest (neighbouring lanes write
// neighbouring 128-byte records)
extern "C" __global__ void __launch_bounds__(256) dvp_weak_anchor_table(const Dev d, const ListArgs a) {
	const int S = d.params.num_images - 1;
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	if (i >= (long long)a.count * S * kAnchors) return;
	const int k = (int)(i % kAnchors);
	const long long r = i / kAnchors;
	const int v0 = (int)(r % S);
	const int t = (int)(r / S);
	// (the records written through an LDS transpose — eight whole records per store instruction instead of 16 bytes into each of
	// 64 lines — measure the same 19 ms per cfg3 pass: the launch does not wait for its stores)
	if (d.images8) build_anchor_record<1>(d, d.weak_list[a.base + t], v0, k);
	else build_anchor_record<0>(d, d.weak_list[a.base + t], v0, k);
}

// replicate the image border into the kImgPad-wide frame of a padded plane set
extern "C" __global__ void dvp_pad_replicate(float* planes, int W, int H, int pitch, size_t plane_stride, int n_planes) {
	const int PW = W + 2 * kImgPad, PH = H + 2 * kImgPad;
	const int frame = 2 * kImgPad * PW + 2 * kImgPad * H;   // border cells per plane
	const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (long long)frame * n_planes) return;
	const int pl = (int)(t / frame);
	int k = (int)(t - (long long)pl * frame);
	int x, y;
	if (k < 2 * kImgPad * PW) {           // top and bottom bands
		y = k / PW;
		x = k - y * PW;
		if (y >= kImgPad) y += H;
	} else {                              // left and right bands of the interior rows
		k -= 2 * kImgPad * PW;
		y = kImgPad + k / (2 * kImgPad);
		x = k % (2 * kImgPad);
		if (x >= kImgPad) x += W;
	}
	float* p = planes + (size_t)pl * plane_stride;
	const int sx = clampi(x - kImgPad, 0, W - 1) + kImgPad, sy = clampi(y - kImgPad, 0, H - 1) + kImgPad;
	p[(size_t)y * pitch + x] = p[(size_t)sy * pitch + sx];
}

// plain padded planes -> row-pair planes: out[(y*pitch + x)*2 + {0,1}] = {in[y][x], in[y+1][x]}
// (the last padded row pairs with itself; the sampler never reads its second component)
extern "C" __global__ void dvp_interleave_rows(const float* __restrict__ in, float* __restrict__ out, int PH, int pitch, size_t plane_stride, int n_planes) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;
	const int pl = blockIdx.z;
	if (x >= pitch) return;
	const float* p = in + (size_t)pl * plane_stride;
	const int y1 = y + 1 < PH ? y + 1 : y;
	const float2 v = make_float2(p[(size_t)y * pitch + x], p[(size_t)y1 * pitch + x]);
	reinterpret_cast<float2*>(out + (size_t)pl * plane_stride * 2)[(size_t)y * pitch + x] = v;
}

// row-pair float planes -> tiled byte planes (Dev::images8); *inexact is raised when a texel is not an integer in [0, 255].
// One thread per tile element: tile (tx, ty), element (ex, ey) = padded pixel (tx*7 + ex, ty*8 + ey), clamped to the
// padded plane (the right-most column of a tile repeats the first pixel of the next tile).
extern "C" __global__ void dvp_pairs_to_tiles(const float* __restrict__ pairs, uint8_t* __restrict__ out, int PW, int PH, int pitch, size_t plane_stride,
                                              int tiles_x, int tiles_y, int n_planes, int* inexact) {
	const size_t per_plane = (size_t)tiles_x * tiles_y * 64;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= per_plane * n_planes) return;
	const int pl = (int)(i / per_plane);
	const size_t r = i - (size_t)pl * per_plane;
	const int tile = (int)(r >> 6), e = (int)(r & 63);
	const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
	const int sx = min(tx * kT8W + (e % kT8E), PW - 1), sy = min(ty * kT8H + (e / kT8E), PH - 1);
	if (e >= kT8E * kT8H) return;
	const float* plane = pairs + (size_t)pl * plane_stride * 2;
	const float2 v = reinterpret_cast<const float2*>(plane)[(size_t)sy * pitch + sx];
	const float2 u = reinterpret_cast<const float2*>(plane)[(size_t)sy * pitch + min(sx + 1, PW - 1)];
	const bool ok = v.x >= 0.0f && v.x <= 255.0f && v.y >= 0.0f && v.y <= 255.0f && v.x == floorf(v.x) && v.y == floorf(v.y);
	if (!ok) { if (*inexact == 0) atomicOr(inexact, 1); return; }
	uint8_t* dst = out + ((size_t)pl * tiles_x * tiles_y + tile) * 128 + (size_t)e * kT8B;
	dst[0] = (uint8_t)v.x;
	dst[1] = (uint8_t)v.y;
	if (kT8Quad) { dst[2] = (uint8_t)u.x; dst[3] = (uint8_t)u.y; }
}

// byte edge map -> 32x32 bit tiles (one thread per 32-bit word)
extern "C" __global__ void dvp_pack_edge_bits(const uint8_t* __restrict__ edge, uint32_t* __restrict__ bits, int W, int H, int tiles_x, size_t words, int equals) {
	const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w < words) bits[w] = pack_edge_word(edge, W, H, tiles_x, w, equals);
}

// visibility-prior candidates of GenEdgeInform: pixels x source views (blockIdx.y = view, wave-uniform)
extern "C" __global__ void __launch_bounds__(256) dvp_gen_candidates(const Dev d, const LaunchArgs a) {
	int px, py;
	if (block_to_pixel(blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py))
		gen_candidates_px(d, px, py, (int)blockIdx.y);
}

// all views of a pixel by one lane (gen_candidates_views_px: the tap weights once instead of once per view)
extern "C" __global__ void __launch_bounds__(256) dvp_gen_candidates_views(const Dev d, const LaunchArgs a) {
	int px, py;
	if (block_to_pixel(blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py))
		gen_candidates_views_px<kCandGroup>(d, px, py, (int)blockIdx.y * kCandGroup);
}
extern "C" __global__ void __launch_bounds__(256) dvp_gen_candidates_views_list(const Dev d, const unsigned* list, const unsigned* n_list) {
	const unsigned t = blockIdx.x * 256 + threadIdx.x;
	if (t >= *n_list) return;
	const int center = (int)list[t];
	const int py = center / d.width, px = center - py * d.width;
	gen_candidates_views_px<kCandGroup>(d, px, py, (int)blockIdx.y * kCandGroup);
}

// ... and the same for the ANCHOR pixels only (dvp_run_patchmatch).  The weak update is the records' one reader, and it reads
// them at the anchors of WEAK pixels (make_anchor_record, wave_ncc_new: the anchor's eight offsets per view, APD.cu:938-952) —
// after GenNeighbours that is a few percent of the image where the reference's GenEdgeInform fills every pixel x view (52 ms
// of side-stream work at 6208x4128, S = 9).  Three launches: mark the anchors of the WEAK list, compact the marks into a
// list (any order), one lane per (listed pixel, view).  They read the selected-view map as GenEdgeInform saw it (a snapshot
// taken before RandomInitialization rewrites it), so the records are those of the full launch.
extern "C" __global__ void __launch_bounds__(256) dvp_anchor_mask(const Dev d, const ListArgs a, uint8_t* mask) {
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t >= a.count) return;
	const int center = d.weak_list[a.base + t];
	const s2* nb = d.neighbours + (size_t)d.neighbours_map[center] * DVP_NEIGHBOUR_NUM;
	for (int k = 1; k < DVP_NEIGHBOUR_NUM; ++k) {
		const s2 q = nb[k];
		if (q.x >= 0 && q.y >= 0 && q.x < d.width && q.y < d.height) mask[(size_t)q.y * d.width + q.x] = 1;
	}
}
extern "C" __global__ void __launch_bounds__(256) dvp_mask_compact(const uint8_t* mask, size_t L, unsigned* list, unsigned* n_list) {
	const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
	const bool go = p < L && mask[p] != 0;
	const unsigned long long m = __ballot(go);
	if (!m) return;
	const int lane = threadIdx.x & 63;
	unsigned base = 0;
	if (lane == __builtin_ctzll(m)) base = atomicAdd(n_list, (unsigned)__popcll(m));
	base = __shfl(base, __builtin_ctzll(m), 64);
	if (go) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)p;
}
extern "C" __global__ void __launch_bounds__(256) dvp_gen_candidates_list(const Dev d, const unsigned* list, const unsigned* n_list) {
	const unsigned t = blockIdx.x * 256 + threadIdx.x;
	if (t >= *n_list) return;
	const int center = (int)list[t];
	const int py = center / d.width, px = center - py * d.width;
	gen_candidates_px(d, px, py, (int)blockIdx.y);
}

extern "C" __global__ void dvp_pack_bits_transposed(const uint8_t* __restrict__ map, uint32_t* __restrict__ bits, int W, int H, int tiles_x, size_t words, int equals) {
	const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w < words) bits[w] = pack_edge_word_t(map, W, H, tiles_x, w, equals);
}

// cell table of the edge map (edge_count_upper): counts per 8x8 cell from the bit tiles, then the two prefix passes
extern "C" __global__ void dvp_edge_cell_counts(const Dev d) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= d.sat_cells_x * d.sat_cells_y) return;
	const int cy = c / d.sat_cells_x, cx = c - cy * d.sat_cells_x;
	int n = 0;
	for (int r = 0; r < 8; ++r) {
		const int y = cy * 8 + r;
		if (y >= d.height) break;
		const unsigned w = d.edge_bits[(size_t)(((y >> 5) * d.edge_tiles_x + (cx >> 2)) * 32 + (y & 31))];
		n += __popc((w >> ((cx & 3) * 8)) & 255u);
	}
	d.edge_sat[(cy + 1) * (d.sat_cells_x + 1) + cx + 1] = n;
}
extern "C" __global__ void dvp_edge_sat_rows(const Dev d) {   // one thread per table row: running sum along x
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r > d.sat_cells_y) return;
	int* row = d.edge_sat + (size_t)r * (d.sat_cells_x + 1);
	int acc = 0;
	for (int x = 0; x <= d.sat_cells_x; ++x) { acc += (r == 0 || x == 0) ? 0 : row[x]; row[x] = acc; }
}
extern "C" __global__ void dvp_edge_sat_cols(const Dev d) {   // one thread per table column: running sum along y
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x > d.sat_cells_x) return;
	const int P = d.sat_cells_x + 1;
	int acc = 0;
	for (int r = 0; r <= d.sat_cells_y; ++r) { acc += d.edge_sat[(size_t)r * P + x]; d.edge_sat[(size_t)r * P + x] = acc; }
}

// sample search of the strong update (same red/black launch geometry, no LDS, small register footprint)
extern "C" __global__ void __launch_bounds__(256) dvp_strong_search(const Dev d, const LaunchArgs a) {
	int px, py;
	if (block_to_pixel(blockIdx.x, threadIdx.x & 63, threadIdx.x >> 6, a.tiles_x, a.tiles, a.rows, a.half, a.colour, d.width, d.height, &px, &py))
		strong_search_px(d, px, py);
}

// line-scan pre-pass of GenEdgeInform: nearest edge pixel in 8 directions (blockIdx.y = direction)
extern "C" __global__ void __launch_bounds__(256) dvp_edge_rays(const Dev d, int what) {
	edge_ray_line(d, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, what);
}

// ---- WEAK-pixel bookkeeping on the device (dvp_upload_state) -------------------------------------------------
// neighbours_map (running index of WEAK pixels in raster order, APD.cpp:1182-1193) and the compacted WEAK list of the
// list kernels used to be built by serial host loops over all L pixels per pass (25.6 M at full resolution: 0.2 s, plus a
// 26 MB download when the weak map was already on the device).  Three small launches instead: ballot counts per unit,
// one exclusive scan, ballot-rank scatter.
// List order: 64 x 64 super-tiles row-major over the image, 16 x 16 tiles row-major inside them, rows inside those; black
// ((x + y) even) pixels first, then red.  "Slot" = (super-tile, tile position 0..15); a slot outside the image counts zero.
constexpr int kWeakTile = 16, kWeakSuper = 64, kWeakChunk = 1024;   // raster chunk of the neighbours_map scan
DVP_HD void weak_slot_origin(int slot, int supers_x, int* tx, int* ty) {
	const int super = slot >> 4, t = slot & 15;
	*tx = (super % supers_x) * kWeakSuper + (t & 3) * kWeakTile;
	*ty = (super / supers_x) * kWeakSuper + (t >> 2) * kWeakTile;
}
// one wave per slot: counts[slot] = black WEAK pixels, counts[n_slots + slot] = red ones;
// one wave per raster chunk: counts[2 * n_slots + chunk] = WEAK pixels of the chunk
extern "C" __global__ void __launch_bounds__(256) dvp_weak_counts(const uint8_t* __restrict__ weak, int W, int H, int supers_x, int n_slots, int n_chunks, int* __restrict__ counts) {
	const int lane = threadIdx.x & 63;
	const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (unit < n_slots) {
		int tx, ty;
		weak_slot_origin(unit, supers_x, &tx, &ty);
		int nb = 0, nr = 0;
		for (int step = 0; step < 4; ++step) {
			const int x = tx + (lane & 15), y = ty + step * 4 + (lane >> 4);
			const bool wk = x < W && y < H && weak[(size_t)y * W + x] == DVP_WEAK;
			nb += __popcll(__ballot(wk && !((x + y) & 1)));
			nr += __popcll(__ballot(wk && ((x + y) & 1)));
		}
		if (lane == 0) { counts[unit] = nb; counts[n_slots + unit] = nr; }
	} else if (unit < n_slots + n_chunks) {
		const int chunk = unit - n_slots;
		const size_t L = (size_t)W * H;
		int n = 0;
		for (int step = 0; step < kWeakChunk / 64; ++step) {
			const size_t i = (size_t)chunk * kWeakChunk + step * 64 + lane;
			n += __popcll(__ballot(i < L && weak[i] == DVP_WEAK));
		}
		if (lane == 0) counts[2 * n_slots + chunk] = n;
	}
}
// exclusive scan of three int arrays in place (one workgroup each): blockIdx.x selects [start[b], start[b + 1]); totals[b] = sum
extern "C" __global__ void __launch_bounds__(1024) dvp_scan3(int* __restrict__ data, int s0, int s1, int s2, int s3, int* __restrict__ totals) {
	__shared__ int part[1024];
	__shared__ int carry;
	const int starts[4] = { s0, s1, s2, s3 };
	const int lo = starts[blockIdx.x], hi = starts[blockIdx.x + 1];
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = lo; base < hi; base += 1024) {
		const int i = base + (int)threadIdx.x;
		const int v = i < hi ? data[i] : 0;
		part[threadIdx.x] = v;
		__syncthreads();
		for (int off = 1; off < 1024; off <<= 1) {
			const int add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
			__syncthreads();
			part[threadIdx.x] += add;
			__syncthreads();
		}
		if (i < hi) data[i] = carry + part[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry += part[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}
// scatter: the list entries of a slot / the running indices of a chunk, in pixel order, by ballot rank
extern "C" __global__ void __launch_bounds__(256) dvp_weak_fill(const uint8_t* __restrict__ weak, int W, int H, int supers_x, int n_slots, int n_chunks,
                                                               const int* __restrict__ offs, const int* __restrict__ totals, int* __restrict__ list, int* __restrict__ map) {
	const int lane = threadIdx.x & 63;
	const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
	const unsigned long long below = (1ull << lane) - 1ull;
	if (unit < n_slots) {
		int tx, ty;
		weak_slot_origin(unit, supers_x, &tx, &ty);
		int ob = offs[unit], orr = totals[0] + offs[n_slots + unit];   // red entries follow all black ones
		for (int step = 0; step < 4; ++step) {
			const int x = tx + (lane & 15), y = ty + step * 4 + (lane >> 4);
			const bool wk = x < W && y < H && weak[(size_t)y * W + x] == DVP_WEAK;
			const bool red = (x + y) & 1;
			const unsigned long long mb = __ballot(wk && !red), mr = __ballot(wk && red);
			if (wk) list[red ? orr + __popcll(mr & below) : ob + __popcll(mb & below)] = y * W + x;
			ob += __popcll(mb);
			orr += __popcll(mr);
		}
	} else if (unit < n_slots + n_chunks) {
		const int chunk = unit - n_slots;
		const size_t L = (size_t)W * H;
		int o = offs[2 * n_slots + chunk];
		for (int step = 0; step < kWeakChunk / 64; ++step) {
			const size_t i = (size_t)chunk * kWeakChunk + step * 64 + lane;
			const bool wk = i < L && weak[i] == DVP_WEAK;
			const unsigned long long m = __ballot(wk);
			if (i < L) map[i] = wk ? o + __popcll(m & below) : 0;
			o += __popcll(m);
		}
	}
}

// RescaleMatToTargetSize (APD.cpp:1773-1795) for the five maps a REFINE_INIT pass inherits from the coarser pyramid level
// (APD.cpp:1428-1456 depth + normal -> planes, selected views, :1169-1181 pixel states, :1648-1667 radius map): nearest
// neighbour with the source's index rule — the ROW index is divided by the WIDTH ratio and the column index by the height
// ratio (`o_r = r / scale_x`, `o_c = c / scale_y`, :1787-1788; the same unless the aspect ratios differ) — one IEEE float
// division and a truncation each; a pixel whose source index falls outside the coarse map stays zero.  Then the radius
// rule of APD.cpp:1660-1666: a pixel whose state is UNKNOWN restarts with the default patch radius.
struct RescaleArgs {
	int W, H, sw, sh;
	float scale_x, scale_y;
	const float* depth; const float* normal; const uint32_t* views; const uint8_t* weak; const int* radius;   // coarse maps; weak / radius may be null
	f4* planes; uint32_t* out_views; uint8_t* out_weak; int* out_radius;
	int radius_fallback;
};
extern "C" __global__ void __launch_bounds__(256) dvp_rescale_state(const RescaleArgs a) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
	if (c >= a.W) return;
	const int o_r = static_cast<int>(r / a.scale_x);
	const int o_c = static_cast<int>(c / a.scale_y);
	const bool in = o_r >= 0 && o_c >= 0 && o_r < a.sh && o_c < a.sw;
	const size_t o = (size_t)r * a.W + c, i = in ? (size_t)o_r * a.sw + o_c : 0;
	a.planes[o] = in ? mk4(a.normal[3 * i], a.normal[3 * i + 1], a.normal[3 * i + 2], a.depth[i]) : mk4(0.0f, 0.0f, 0.0f, 0.0f);
	a.out_views[o] = in ? a.views[i] : 0u;
	const uint8_t st = a.weak ? (in ? a.weak[i] : (uint8_t)0) : (uint8_t)DVP_STRONG;
	a.out_weak[o] = st;
	if (a.radius) a.out_radius[o] = st == DVP_UNKNOWN ? a.radius_fallback : (in ? a.radius[i] : 0);
}

// What the driver makes of the downloaded planes (main.cpp:300-309): depth map = plane.w where it lies inside
// [depth_min, depth_max], else 0 and the pixel's state becomes UNKNOWN; normal map = plane.xyz.  (`!(w < min || w > max)`
// as the source writes it: a NaN depth is kept.)
extern "C" __global__ void __launch_bounds__(256) dvp_unpack_maps(const f4* __restrict__ planes, const uint8_t* __restrict__ weak, size_t L, float dmin, float dmax,
                                                                 float* __restrict__ depth, float* __restrict__ normal, uint8_t* __restrict__ state) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= L) return;
	const f4 ph = planes[i];
	const bool usable = !(ph.w < dmin || ph.w > dmax);
	depth[i] = usable ? ph.w : 0.0f;
	normal[3 * i] = ph.x; normal[3 * i + 1] = ph.y; normal[3 * i + 2] = ph.z;
	state[i] = usable ? weak[i] : (uint8_t)DVP_UNKNOWN;
}

extern "C" __global__ void dvp_prepare_views(const DvpCamera* cams, ViewConst* views, int n) {
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v >= 1 && v < n) compute_view_const(cams[0], cams[v], &views[v]);
}

// ComputeMultiViewCostVectorOld over a list of (pixel, plane) pairs — KATs and the roofline
// micro-benchmark.  One lane per pair.
extern "C" __global__ void __launch_bounds__(256) dvp_cost_vectors(const Dev d, const int* px, const f4* planes, int n, float* out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int x = px[2 * i], y = px[2 * i + 1];
	const int S = d.num_images - 1;
	PatchCtx c;
	__shared__ f2 lds_tab[kTaps * kTaps * 256];
	int radius, inc;
	patch_geometry(d, x + y * d.width, &radius, &inc);
	build_patch_ctx(d, x, y, radius, inc, 0, PatchTab{&lds_tab[threadIdx.x], 256}, &c);
	const f4 pl = planes[i];
	for (int v = 0; v < S; ++v) out[(size_t)i * S + v] = d.sampler ? ncc_old<1>(d, c, x, y, v + 1, pl) : ncc_old<0>(d, c, x, y, v + 1, pl);
}

// same computation on every pixel with its current plane (camera frame); writes the view-mean
extern "C" __global__ void __launch_bounds__(256) dvp_cost_all_pixels(const Dev d, const LaunchArgs a, float* out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int px, py;
	if (!block_to_pixel(blockIdx.x, lane, wave, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py)) return;
	const int center = px + py * d.width;
	const int S = d.num_images - 1;
	PatchCtx c;
	__shared__ f2 lds_tab[kTaps * kTaps * 256];
	int radius, inc;
	patch_geometry(d, center, &radius, &inc);
	build_patch_ctx(d, px, py, radius, inc, 0, PatchTab{&lds_tab[threadIdx.x], 256}, &c);
	const f4 pl = d.planes[center];
	float acc = 0.0f;
	for (int v = 0; v < S; ++v) acc += ncc_old<0>(d, c, px, py, v + 1, pl);
	out[center] = acc / S;
}


#ifdef DVP_PROBE
// ---- mapping probe (experiment, not part of the product): K evaluations x S views per pixel with the planes of pixels
// `stride` apart, (A) lane = pixel like the NCC kernels, (B) lane = (pixel, view): 64 / S x-adjacent pixels per wave, the
// patch table of a pixel shared by its S lanes in LDS.
#ifndef DVP_PROBE_LB
#define DVP_PROBE_LB 2
#endif
extern "C" __global__ void __launch_bounds__(256, 2) dvp_probe_a(const Dev d, const LaunchArgs a, float* out, int K, int stride) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int px, py;
	if (!block_to_pixel(blockIdx.x, lane, wave, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py)) return;
	const int center = px + py * d.width;
	const int S = d.num_images - 1, L = d.width * d.height;
	PatchCtx c;
	__shared__ f2 lds_tab[kTaps * kTaps * 256];
	int radius, inc;
	patch_geometry(d, center, &radius, &inc);
	build_patch_ctx(d, px, py, radius, inc, 0, PatchTab{&lds_tab[threadIdx.x], 256}, &c);
	float acc = 0.0f;
	for (int k = 0; k < K; ++k) {
		int q = center + (k - K / 2) * stride;
		q = q < 0 ? 0 : (q >= L ? L - 1 : q);
		const f4 pl = d.planes[q];
		for (int v = 0; v < S; ++v) acc += ncc_old<0>(d, c, px, py, v + 1, pl);
	}
	out[center] = acc;
}
extern "C" __global__ void __launch_bounds__(256, DVP_PROBE_LB) dvp_probe_b(const Dev d, float* out, int K, int stride) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int S = d.num_images - 1, L = d.width * d.height;
	const int P = 64 / S;                                // pixels per wave
	const int waves_x = (d.width + P - 1) / P;
	const int gw = blockIdx.x * 4 + wave;
	const int py = gw / waves_x, x0 = (gw - py * waves_x) * P;
	if (py >= d.height) return;
	__shared__ f2 tab[4][9 * 36];                         // [wave][pixel][tap] (P <= 9... here P * 36 <= 324 for S >= 7)
	__shared__ float caa[4][9 * 36];
	__shared__ float sums[4][9][3];
	const int v = lane / P, pi = lane - v * P;           // lanes of one view are contiguous
	const int px = x0 + pi;
	const bool active = v < S && px < d.width;
	const int radius = d.params.strong_radius, inc = d.params.strong_increment;
	// table: entry e = pixel * 36 + tap, built by whichever lane
	for (int e = lane; e < P * 36; e += 64) {
		const int pe = e / 36, t = e - pe * 36, ty = t / 6, tx = t - ty * 6;
		const int qx = x0 + pe < d.width ? x0 + pe : d.width - 1;
		const float cpix = img_texel(d.images, d.org, d.pitch, d.width, d.height, qx, py);
		const float av = img_texel(d.images, d.org, d.pitch, d.width, d.height, qx - radius + tx * inc, py - radius + ty * inc);
		const float w = bilateral_weight((float)(-radius + tx * inc), (float)(-radius + ty * inc), av, cpix, d.params.sigma_spatial, d.params.sigma_color, 0);
		tab[wave][e] = mk2(w, w * av);
		caa[wave][e] = w * av * av;
	}
	__builtin_amdgcn_wave_barrier();
	if (lane < P) {
		float sr = 0.0f, srr = 0.0f, ws = 0.0f;
		for (int ty = 0; ty < 6; ++ty) {
			float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
			for (int tx = 0; tx < 6; ++tx) { const f2 t = tab[wave][lane * 36 + ty * 6 + tx]; a0 += t.y; a1 += caa[wave][lane * 36 + ty * 6 + tx]; a2 += t.x; }
			sr += a0; srr += a1; ws += a2;
		}
		sums[wave][lane][0] = sr; sums[wave][lane][1] = srr; sums[wave][lane][2] = ws;
	}
	__builtin_amdgcn_wave_barrier();
	if (!active) return;
	PatchCtx c;
	c.tab = PatchTab{&tab[wave][pi * 36], 1};
	c.sum_ref = sums[wave][pi][0]; c.sum_ref_ref = sums[wave][pi][1]; c.wsum = sums[wave][pi][2];
	c.radius = radius; c.inc = inc; c.fast = 1;
	const ViewConst vc = d.views[v + 1];                  // per-lane record (vector loads)
	const unsigned plane_off = (unsigned)((size_t)(v + 1) * d.plane_stride * 2 * sizeof(float));
	const int center = px + py * d.width;
	float acc = 0.0f;
	for (int k = 0; k < K; ++k) {
		int q = center + (k - K / 2) * stride;
		q = q < 0 ? 0 : (q >= L ? L - 1 : q);
		const f4 pl = d.planes[q];
		float H[9];
		homography(vc, pl, H);
		const f2 pt = apply_homography(H, px, py);
		if (pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f) { acc += 2.0f; continue; }
		acc += ncc_patch_fast<0>(d, c, H, d.images, px, py, plane_off);
	}
	atomicAdd(&out[center], acc);
}
// ---- round 6: the SWEEP-shaped probe.  Work per pixel = DepthToWeak's: the 61 disparity slots around the
// pixel's own plane + the current depth, against the views of `vmask`.  (C) lane = pixel, slots in a loop — the mapping of
// dvp_sweep_eval; (D) wave = ONE pixel, lane = slot: the patch table is wave-uniform (288 B, read by LDS broadcast), a tap's 62
// gathers walk one epipolar segment.  Both sum a pixel's costs views-inside-slots in the same order: identical bits.
__device__ __forceinline__ float probe_slot_plane(const DvpCamera& rc, const f4 origin, float base_line, float disp, int k, int px, int py, const DvpParams& P, bool* in_range) {
	float p_depth = origin.w;
	*in_range = true;
	if (k < 61) {
		p_depth = rc.K[0] * base_line / (disp + (k - 30));
		if (p_depth < P.depth_min || p_depth > P.depth_max) *in_range = false;
	}
	return distance_to_origin(rc, px, py, p_depth, origin);
}
extern "C" __global__ void __launch_bounds__(256, 2) dvp_probe_c(const Dev d, const LaunchArgs a, float* out, unsigned vmask) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int px, py;
	if (!block_to_pixel(blockIdx.x, lane, wave, a.tiles_x, a.tiles, a.rows, 0, 0, d.width, d.height, &px, &py)) return;
	const int center = px + py * d.width;
	const int S = d.num_images - 1;
	PatchCtx c;
	__shared__ f2 lds_tab[kTaps * kTaps * 256];
	int radius, inc;
	patch_geometry(d, center, &radius, &inc);
	build_patch_ctx(d, px, py, radius, inc, 0, PatchTab{&lds_tab[threadIdx.x], 256}, &c);
	const DvpCamera rc = load_camera(d, 0);
	const f4 origin = normal_world_to_cam(rc, d.planes[center]);
	const float base_line = 0.4f, disp = rc.K[0] * base_line / origin.w;
	float acc = 0.0f;
	for (int k = 0; k < 62; ++k) {
		bool ok;
		f4 pl = origin;
		pl.w = probe_slot_plane(rc, origin, base_line, disp, k, px, py, d.params, &ok);
		float t = 0.0f;
		if (ok) for (int v = 0; v < S; ++v) if ((vmask >> v) & 1) t += ncc_old<0>(d, c, px, py, v + 1, pl);
		acc += t;
	}
	out[center] = acc;
}
#ifndef DVP_PROBE_NPX
#define DVP_PROBE_NPX 16
#endif
extern "C" __global__ void __launch_bounds__(64, DVP_PROBE_LB) dvp_probe_d(const Dev d, float* out, unsigned vmask) {
	const int lane = threadIdx.x;
	const int S = d.num_images - 1;
	__shared__ f2 ctab[kTaps * kTaps];
	__shared__ float caa[kTaps * kTaps];
	__shared__ float tot[64];
	// block -> run of DVP_PROBE_NPX pixels of one row: XCD b % 8 owns a 64-pixel column of a strip of 8, four runs per row, rows top to bottom
	const int runs = 64 / DVP_PROBE_NPX;
	const int tiles_x = (d.width + 63) / 64;
	const int per_strip = 8 * d.height * runs;
	const int st = (int)blockIdx.x / per_strip, rem = (int)blockIdx.x - st * per_strip;
	const int w_last = tiles_x - st * 8 >= 8 ? 8 : tiles_x - st * 8;
	const int col = rem % w_last, k2 = rem / w_last;
	const int py = k2 / runs, x0 = (st * 8 + col) * 64 + (k2 - py * runs) * DVP_PROBE_NPX;
	if (py >= d.height) return;
	const DvpCamera rc = load_camera(d, 0);
	for (int i = 0; i < DVP_PROBE_NPX; ++i) {
		const int px = x0 + i;
		if (px >= d.width) break;
		const int center = px + py * d.width;
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		PatchCtx c;
		c.radius = radius; c.inc = inc; c.fast = 1;
		c.tab = PatchTab{ctab, 1};
		if (lane < kTaps * kTaps) {
			const int ty = lane / kTaps, tx = lane - ty * kTaps;
			const float cpix = img_texel(d.images, d.org, d.pitch, d.width, d.height, px, py);
			const float av = img_texel(d.images, d.org, d.pitch, d.width, d.height, px - radius + tx * inc, py - radius + ty * inc);
			const float w = bilateral_weight((float)(-radius + tx * inc), (float)(-radius + ty * inc), av, cpix, d.params.sigma_spatial, d.params.sigma_color, 0);
			const float wa = w * av;
			ctab[lane] = mk2(w, wa);
			caa[lane] = wa * av;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		{
			float sr = 0.0f, srr = 0.0f, ws = 0.0f;
			for (int ty = 0; ty < kTaps; ++ty) {
				float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
				for (int tx = 0; tx < kTaps; ++tx) { const f2 t = ctab[ty * kTaps + tx]; a0 += t.y; a1 += caa[ty * kTaps + tx]; a2 += t.x; }
				sr += a0; srr += a1; ws += a2;
			}
			c.sum_ref = sr; c.sum_ref_ref = srr; c.wsum = ws;
		}
		f4 origin = normal_world_to_cam(rc, d.planes[center]);
		origin.x = uniform_f(origin.x); origin.y = uniform_f(origin.y); origin.z = uniform_f(origin.z); origin.w = uniform_f(origin.w);
		const float base_line = 0.4f, disp = rc.K[0] * base_line / origin.w;
		bool ok;
		f4 pl = origin;
		pl.w = probe_slot_plane(rc, origin, base_line, disp, lane, px, py, d.params, &ok);
		float t = 0.0f;
		if (lane < 62 && ok) for (int v = 0; v < S; ++v) if ((vmask >> v) & 1) t += ncc_old<0>(d, c, px, py, v + 1, pl);
		tot[lane] = t;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (lane == 0) {
			float acc = 0.0f;
			for (int k = 0; k < 62; ++k) acc += tot[k];
			out[center] = acc;
		}
		__builtin_amdgcn_wave_barrier();
	}
}
extern "C" int dvp_probe(dvp_ctx* c, int mode, int K, int stride, int repeat, float* mean_ms, float* checksum);
#endif

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct EventPair { int stage; hipEvent_t a, b; };   // stage: launch site, or one of the two whole-run timers below
enum { EV_TOTAL = -1, EV_ITER_LOOP = -2 };

struct dvp_ctx {
	int device = 0;
	int W = 0, H = 0, NI = 0, pitch = 0;
	size_t L = 0;
	hipStream_t stream = nullptr;
	// second stream of dvp_run_patchmatch: the visibility-prior candidates (VALU-bound, read by the weak updates only)
	// run beside the latency-bound list kernels of the weak path's preparation
	// visibility-prior candidates at anchor pixels only (dvp_run_patchmatch; DVP_CAND_MASK=0: every pixel, as dvp_run_stage)
	// Which form pays depends on the WEAK share: the full launch runs on the side stream beside the latency-bound anchor search and
	// is hidden when that takes long enough (>= ~4.5 % WEAK at 6208x4128: bench cfg3, 6.9 % — masked there it would run beside
	// RandomInit and the strong update instead and cost them 28 ms); at the 1-3 % WEAK of the real schedule's full-size passes the
	// anchor search is over after 12-38 ms and the full launch was 50-60 ms exposed per view (profiles/r06_e2e_apd.txt).
	// DVP_CAND_MASK=1 / 0 forces a form; default: masked below 4 % WEAK.
	int cand_mask_mode = -1;
	bool cand_mask_on = true;
	uint8_t* cand_mask = nullptr;
	unsigned *cand_list = nullptr, *cand_n = nullptr;
	uint32_t* sel_snap = nullptr;
	hipStream_t side = nullptr;
	hipEvent_t side_fork = nullptr, side_join = nullptr;
	Dev d{};
	std::vector<void*> allocs;
	// named device buffers
	float* images = nullptr;        // row-pair planes (dvp_dev.hpp: img_texel / load_quad)
	uint8_t* images8 = nullptr;     // the same as bytes; images8_ok says whether the last upload was 8-bit exact
	int* images8_flag = nullptr;
	bool images8_ok = false;
	bool no_images8 = false;        // DVP_NO_IMAGES8 in the environment: keep the float planes for every kernel (A/B measurements)
	float* image_stage = nullptr;   // plain padded planes the uploads land in before dvp_interleave_rows
	float* depths = nullptr;
	uint32_t* edge_bits = nullptr;  // bit-tiled copy of `edge`, rebuilt before the launches that walk lines
	uint32_t* strong_bits = nullptr; // bit-tiled (weak_info == STRONG), rebuilt before FindNearestStrongPoint and GenNeighbours
	uint32_t* strong_bits_t = nullptr;   // ... with transposed tiles (FindNearestStrongPoint's column segments)
	int* edge_sat = nullptr;         // cell table of the edge map (Dev::edge_sat), rebuilt with edge_bits
	DvpCamera* cameras = nullptr; ViewConst* views = nullptr; int* sector_taps = nullptr; int* sector_start = nullptr;
	f4* planes = nullptr; f4* planes_snap = nullptr; f4* fit_planes = nullptr;
	int* search_pos = nullptr;   // [16][L]
	float* slot_costs = nullptr; // [17][S][half_w * H]: split strong update (allocated at its first launch)
	float* strong_rec = nullptr; // [SR_FIELDS][half_w * H]
	bool strong_split = true;    // DVP_STRONG_SPLIT=0 in the environment: the monolithic kernel (A/B measurements)
	bool eval_items = true;      // DVP_EVAL_ITEMS=0: dvp_strong_eval with a pixel per lane instead of (pixel, slot) items over the lanes
	bool refine_lanes = true;    // DVP_REFINE_LANES=0: dvp_strong_refine with the wave in lock step over hypotheses and views
	f4* sweep_rec = nullptr; float* sweep_cost = nullptr; float* sweep_pc = nullptr;   // DepthToWeak + LocalRefine as view-compacted passes (allocated at the first fused launch)
	bool ransac_wave = false;    // DVP_RANSAC_WAVE=1: RANSACToGetFitPlane one wave per WEAK pixel, lane = draw (round 6: measured no faster, 16.1 vs 15.9 ms at cfg3, 55.6 vs 49.0 at 25 % WEAK)
	bool sweep_split = true;     // DVP_SWEEP_SPLIT=0, or the buffers did not fit: the fused per-pixel kernel
	bool sweep_force = false;    // DVP_SWEEP_SPLIT=2: the passes also without the geometric term (tests)
	double sweep_band_gb = 0.0;  // DVP_SWEEP_BAND_GB=g: sweep_cost holds a band of rows of at most g GB and the passes run band after band (0: the whole image, 67 GB at 6208x4128 with 9 sources)
	int sweep_band_rows = 0;     // rows of a band (a multiple of the evaluation tile's rows); 0 = not banded
	bool gn_wave = false;        // DVP_GN_WAVE=1: GenNeighbours' search as one wave per WEAK pixel (dvp_gen_neighbours_search; measured slower)
	float* costs = nullptr; float* costs_snap = nullptr; float* complex_ = nullptr;
	uint32_t* selected_views = nullptr;
	uint8_t* view_weight = nullptr; uint8_t* weak_info = nullptr; uint8_t* weak_reliable = nullptr; uint8_t* edge = nullptr;
	s2* label_stop = nullptr; s2* weak_nearest_strong = nullptr; s2* neighbours = nullptr; s2* gn_points = nullptr; int* gn_count = nullptr; s2* candidate = nullptr; s2* edge_neigh = nullptr; s2* label_boundary = nullptr;
	int* neighbours_map = nullptr; int* label = nullptr; int* radius = nullptr;
	unsigned long long* eval_counter = nullptr;
	float* scratch_out = nullptr;
	size_t weak_alloc = 0;       // capacity (in WEAK pixels) of the per-WEAK buffers
	AnchorRec* anchor_tab = nullptr;   // [WEAK][S][11] reference sides of the anchor sub-patches (dvp_weak_wave.hpp), allocated at the first weak update
	size_t anchor_tab_alloc = 0;       // capacity in records
	bool anchor_tab_valid = false;     // built for the current anchors / offsets / images (any launch or upload that can change them clears it)
	bool anchor_tab_off = false;       // DVP_WEAK_ANCHOR_TAB=0, or the table did not fit: the weak update forms the reference side per item
	// the weak update as seven launches (dvp_weak_phased.hpp): per-WEAK-pixel hand-over, allocated at the first weak update
	WeakRec* weak_rec = nullptr; f2* weak_ctab = nullptr; float* weak_ev = nullptr;
	size_t weak_phase_alloc = 0;       // capacity in WEAK pixels
	bool weak_phased = true;           // DVP_WEAK_PHASED=0, no anchor table, or the buffers did not fit: the one-wave form
	int weak_phased_min = 8192;        // WEAK pixels of a launch below which the one-wave kernel is used (DVP_WEAK_PHASED_MIN; tests: 0)
	int weak_run[4] = { 64, 256, 1024, 1024 };   // WEAK pixels per XCD run of the same launches (DVP_WEAK_RUNS=a,b,c,d)
	int weak_group[4] = { 1, 4, 4, 2 };   // WEAK pixels per wave of E0 / E1 / E2a / E2b (DVP_WEAK_GROUPS=a,b,c,d: A/B measurements)
	int* weak_list = nullptr;    // compacted WEAK pixel indices (black first, then red)
	size_t weak_list_alloc = 0;
	int* weak_counts = nullptr;  // scratch of the device-side compaction: per-slot black / red counts, per-chunk counts, then 3 totals
	int* weak_totals_host = nullptr;   // pinned: (black, red, all)
	uint8_t* coarse = nullptr;   // staging of the coarser level's maps (dvp_upload_state_rescaled)
	uint8_t* maps_out = nullptr; // staging of dvp_download_maps: depth [L] f32, normal [L][3] f32, states [L] u8, selected views [L] u32, radius [L] i32
	// dvp_download_maps_begin / _finish: the copies to the host run on their own stream, from any thread, while this context is
	// already on its next view; `dl_busy` = maps_out holds maps that have not been fetched yet
	hipStream_t copy = nullptr;
	uint8_t* maps_host = nullptr;   // pinned mirror of maps_out (one DMA, no staging through the runtime's pageable-copy path)
	std::mutex dl_m;
	std::condition_variable dl_cv;
	bool dl_busy = false;
	bool dl_fetching = false;   // a dvp_download_maps_finish call is running right now (it ends in bounded time)
	size_t coarse_alloc = 0;
	// dvp_save_state / dvp_restore_state: device-side copy of the per-pixel input state
	f4* saved_planes = nullptr; uint32_t* saved_views = nullptr; uint8_t* saved_weak = nullptr; int* saved_radius = nullptr;
	bool have_saved = false;
	int lut_radius = -1;
	bool have_depths = false;
	bool profiling = false;
	std::vector<EventPair> events;
	DvpTimings timings{};
	std::string error;
};

static std::string g_create_error;

#define HIP_TRY(ctx, expr)                                                                          \
	do {                                                                                            \
		hipError_t e_ = (expr);                                                                     \
		if (e_ != hipSuccess) {                                                                     \
			char buf_[512];                                                                         \
			snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			(ctx)->error = buf_;                                                                    \
			return 1;                                                                               \
		}                                                                                           \
	} while (0)

template <class T>
static int dalloc(dvp_ctx* c, T** p, size_t count, bool zero = true) {
	void* q = nullptr;
	const size_t bytes = (count ? count : 1) * sizeof(T);
	HIP_TRY(c, hipMalloc(&q, bytes));
	c->allocs.push_back(q);
	if (zero) HIP_TRY(c, hipMemsetAsync(q, 0, bytes, c->stream));
	*p = (T*)q;
	return 0;
}

// give a superseded block back (the caller has made sure no queued work still uses it)
template <class T>
static void dfree(dvp_ctx* c, T** p) {
	if (!*p) return;
	for (size_t i = 0; i < c->allocs.size(); ++i)
		if (c->allocs[i] == (void*)*p) { c->allocs[i] = c->allocs.back(); c->allocs.pop_back(); break; }
	(void)hipFree((void*)*p);
	*p = nullptr;
}

static void sync_dev_struct(dvp_ctx* c) {
	Dev& d = c->d;
	d.width = c->W; d.height = c->H; d.num_images = c->NI; d.pitch = c->pitch;
	d.org = kImgPad * c->pitch + kImgPad;
	d.plane_stride = (size_t)c->pitch * (c->H + 2 * kImgPad);
	d.images = c->images; d.images8 = c->images8_ok ? c->images8 : nullptr; d.img8_tiles_x = img8_tiles_x(c->W); d.img8_plane_bytes = (size_t)img8_tiles_x(c->W) * img8_tiles_y(c->H) * 128; d.depths = c->depths; d.cameras = c->cameras; d.views = c->views; d.sector_taps = c->sector_taps; d.sector_start = c->sector_start;
	d.search_pos = c->search_pos;
	d.sweep_px0 = 0; d.sweep_row0 = 0; d.sweep_row1 = 0;   // (set per band by the sweep passes' launches)
	d.sweep_rec = c->sweep_rec; d.sweep_cost = c->sweep_cost; d.sweep_pc = c->sweep_pc; d.slot_costs = c->slot_costs; d.strong_rec = c->strong_rec; d.half_w = (c->W + 1) / 2;
	d.planes = c->planes; d.planes_snap = c->planes_snap; d.costs = c->costs; d.costs_snap = c->costs_snap;
	d.selected_views = c->selected_views; d.view_weight = c->view_weight; d.weak_info = c->weak_info;
	d.weak_reliable = c->weak_reliable; d.weak_nearest_strong = c->weak_nearest_strong;
	d.anchor_tab = (c->anchor_tab_off || !c->anchor_tab_valid) ? nullptr : c->anchor_tab; d.neighbours_map = c->neighbours_map; d.neighbours = c->neighbours; d.gn_points = c->gn_points; d.gn_count = c->gn_count; d.fit_planes = c->fit_planes;
	d.candidate = c->candidate; d.edge = c->edge; d.edge_bits = c->edge_bits; d.strong_bits = c->strong_bits; d.strong_bits_t = c->strong_bits_t; d.edge_sat = c->edge_sat; d.sat_cells_x = sat_cells(c->W); d.sat_cells_y = sat_cells(c->H); d.edge_tiles_x = edge_tiles_x(c->W); d.edge_neigh = c->edge_neigh; d.label = c->label;
	d.label_boundary = c->label_boundary; d.label_stop = c->label_stop; d.complex_ = c->complex_; d.radius = c->radius;
	d.weak_list = c->weak_list;
	d.weak_rec = c->weak_rec; d.weak_ctab = c->weak_ctab; d.weak_ev = c->weak_ev;
	d.eval_counter = c->profiling ? c->eval_counter : nullptr;
}

static int set_device(dvp_ctx* c) {
	HIP_TRY(c, hipSetDevice(c->device));
	return 0;
}

extern "C" {

int dvp_ctx_create(int device, int width, int height, int num_images, dvp_ctx** out) {
	if (!out) return 1;
	*out = nullptr;
	if (width <= 0 || height <= 0 || num_images < 2 || num_images > DVP_MAX_IMAGES || width > 32767 || height > 32767) {
		g_create_error = "dvp_ctx_create: bad dimensions (2 <= num_images <= 32, sizes <= 32767: short2 pixel coordinates)";
		return 1;
	}
	{   // the samplers address a row-pair plane with 32-bit byte offsets
		const unsigned long long pitch = ((unsigned long long)width + 2 * kImgPad + 63) / 64 * 64;
		if (pitch * ((unsigned long long)height + 2 * kImgPad) * 8ull >= (1ull << 32)) {
			g_create_error = "dvp_ctx_create: image too large (a padded row-pair plane must stay below 4 GiB)";
			return 1;
		}
	}
	dvp_ctx* c = new dvp_ctx();
	c->device = device; c->W = width; c->H = height; c->NI = num_images;
	c->no_images8 = getenv("DVP_NO_IMAGES8") != nullptr;
	if (const char* e = getenv("DVP_STRONG_SPLIT")) c->strong_split = atoi(e) != 0;
	if (const char* e = getenv("DVP_REFINE_LANES")) c->refine_lanes = atoi(e) != 0;
	if (const char* e = getenv("DVP_EVAL_ITEMS")) c->eval_items = atoi(e) != 0;
	if (const char* e = getenv("DVP_SWEEP_SPLIT")) { c->sweep_split = atoi(e) != 0; c->sweep_force = atoi(e) == 2; }
	if (const char* e = getenv("DVP_SWEEP_BAND_GB")) c->sweep_band_gb = atof(e);
	if (const char* e = getenv("DVP_WEAK_ANCHOR_TAB")) c->anchor_tab_off = atoi(e) == 0;   // A/B measurements and the tests of the per-item form
	if (const char* e = getenv("DVP_GN_WAVE")) c->gn_wave = atoi(e) != 0;
	if (const char* e = getenv("DVP_RANSAC_WAVE")) c->ransac_wave = atoi(e) != 0;
	if (const char* e = getenv("DVP_CAND_MASK")) c->cand_mask_mode = atoi(e) != 0 ? 1 : 0;
	if (const char* e = getenv("DVP_WEAK_PHASED")) c->weak_phased = atoi(e) != 0;
	if (const char* e = getenv("DVP_WEAK_PHASED_MIN")) c->weak_phased_min = atoi(e);
	if (const char* e = getenv("DVP_WEAK_RUNS")) {
		int g[4];
		if (sscanf(e, "%d,%d,%d,%d", &g[0], &g[1], &g[2], &g[3]) == 4)
			for (int i = 0; i < 4; ++i) c->weak_run[i] = g[i] < 1 ? 1 : g[i];
	}
	if (const char* e = getenv("DVP_WEAK_GROUPS")) {
		int g[4];
		if (sscanf(e, "%d,%d,%d,%d", &g[0], &g[1], &g[2], &g[3]) == 4)
			for (int i = 0; i < 4; ++i) c->weak_group[i] = g[i] < 1 ? 1 : (g[i] > kGrp ? kGrp : g[i]);   // (E0: at most kGrpWide, applied at the launch)
	}
	c->pitch = (width + 2 * kImgPad + 63) / 64 * 64;
	c->L = (size_t)width * height;
	auto fail = [&](int) { g_create_error = c->error; dvp_ctx_destroy(c); return 1; };
	if (set_device(c)) return fail(0);
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->error = "hipStreamCreate failed"; return fail(0); }
	if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming) != hipSuccess) { c->error = "hipStreamCreate failed"; return fail(0); }
	const size_t L = c->L, S = (size_t)num_images - 1, plane = (size_t)c->pitch * (height + 2 * kImgPad);
	int r = 0;
	r |= dalloc(c, &c->images, plane * num_images * 2);
	r |= dalloc(c, &c->images8, (size_t)img8_tiles_x(width) * img8_tiles_y(height) * 128 * num_images);
	r |= dalloc(c, &c->images8_flag, (size_t)1);
	r |= dalloc(c, &c->image_stage, plane * num_images);
	r |= dalloc(c, &c->cameras, (size_t)num_images);
	r |= dalloc(c, &c->views, (size_t)num_images);
	r |= dalloc(c, &c->planes, L);
	r |= dalloc(c, &c->planes_snap, L);
	r |= dalloc(c, &c->search_pos, L * 16);
	r |= dalloc(c, &c->fit_planes, L);                     // cudaMemset 0, APD.cpp:1571
	r |= dalloc(c, &c->costs, L);
	r |= dalloc(c, &c->costs_snap, L);
	r |= dalloc(c, &c->selected_views, L + (size_t)width); // + one zeroed row (APD.cu:2473)
	r |= dalloc(c, &c->view_weight, L * 32);
	r |= dalloc(c, &c->weak_info, L);
	r |= dalloc(c, &c->weak_reliable, L);
	r |= dalloc(c, &c->weak_nearest_strong, L);
	r |= dalloc(c, &c->neighbours_map, L);
	r |= dalloc(c, &c->candidate, L * S * 8);
	r |= dalloc(c, &c->edge, L);
	r |= dalloc(c, &c->edge_bits, edge_bits_words(width, height));
	r |= dalloc(c, &c->strong_bits, edge_bits_words(width, height));
	r |= dalloc(c, &c->strong_bits_t, edge_bits_words(width, height));
	r |= dalloc(c, &c->edge_sat, (size_t)(sat_cells(width) + 1) * (sat_cells(height) + 1));
	r |= dalloc(c, &c->edge_neigh, L * 8);
	r |= dalloc(c, &c->label, L);
	r |= dalloc(c, &c->radius, L);
	r |= dalloc(c, &c->eval_counter, (size_t)1);
	r |= dalloc(c, &c->scratch_out, L);
	if (r) return fail(0);
	// defaults: every pixel STRONG (APD.cpp:1196-1204), radius = strong_radius (APD.cpp:1649-1653)
	{
		std::vector<uint8_t> st(L, (uint8_t)DVP_STRONG);
		std::vector<int> rad(L, 5);
		std::vector<s2> m1(L * 8, mks2(-1, -1));
		if (hipMemcpyAsync(c->weak_info, st.data(), L, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			hipMemcpyAsync(c->radius, rad.data(), L * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			hipMemcpyAsync(c->edge_neigh, m1.data(), L * 8 * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			hipMemcpyAsync(c->weak_nearest_strong, m1.data(), L * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
			hipStreamSynchronize(c->stream) != hipSuccess) { c->error = "default state upload failed"; return fail(0); }
	}
	std::memset(&c->d.params, 0, sizeof(DvpParams));
	c->d.params.num_images = num_images;
	c->d.params.strong_radius = 5; c->d.params.strong_increment = 2; c->d.params.weak_radius = 5; c->d.params.weak_increment = 5;
	c->d.params.rotate_time = 4;
	sync_dev_struct(c);
	*out = c;
	return 0;
}

int dvp_ctx_destroy(dvp_ctx* c) {
	if (!c) return 0;
	// teardown: errors are not actionable here
	(void)hipSetDevice(c->device);
	{   // maps another thread is still fetching (dvp_download_maps_finish)
		// The driver relies on this wait: it hands a context back (and may destroy it) while the view's background job has
		// not fetched the staged maps yet.  Bounded all the same: maps that were staged and are never fetched — an exception
		// between the two steps — count as abandoned after DVP_DOWNLOAD_WAIT_S (120) seconds; a fetch that is RUNNING is
		// always waited for.
		int limit_s = 120;
		if (const char* e = getenv("DVP_DOWNLOAD_WAIT_S")) limit_s = atoi(e);
		std::unique_lock<std::mutex> lk(c->dl_m);
		c->dl_cv.wait_for(lk, std::chrono::seconds(limit_s > 0 ? limit_s : 1), [c] { return !c->dl_busy; });
		c->dl_cv.wait(lk, [c] { return !c->dl_fetching; });
	}
	if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
	if (c->maps_host) (void)hipHostFree(c->maps_host);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	for (auto& e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	for (void* p : c->allocs) (void)hipFree(p);
	if (c->weak_totals_host) (void)hipHostFree(c->weak_totals_host);
	if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
	if (c->side_fork) (void)hipEventDestroy(c->side_fork);
	if (c->side_join) (void)hipEventDestroy(c->side_join);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return 0;
}

#ifndef DVP_BUILD_ID
#define DVP_BUILD_ID "unknown"
#endif
const char* dvp_build_id(void) { return DVP_BUILD_ID; }

// dvp_download_maps_finish may run on another thread than the one that drives the context: its error text stays with
// the calling thread instead of racing with the driver thread's writes to dvp_ctx::error
struct FinishError { const dvp_ctx* c = nullptr; std::string msg; };
static thread_local FinishError t_finish_error;
const char* dvp_last_error(const dvp_ctx* c) {
	if (c && t_finish_error.c == c && !t_finish_error.msg.empty()) return t_finish_error.msg.c_str();
	return c ? c->error.c_str() : g_create_error.c_str();
}

// `pairs` != nullptr: `dst` is the staging set and the row-pair planes are produced from it
static int upload_planes(dvp_ctx* c, float* dst, const float* const* src, int pitch_floats, hipMemcpyKind kind, float* pairs = nullptr) {
	if (set_device(c)) return 1;
	if (pitch_floats < c->W) { c->error = "pitch_floats < width"; return 1; }
	const size_t stride = (size_t)c->pitch * (c->H + 2 * kImgPad);
	for (int i = 0; i < c->NI; ++i) {
		if (!src[i]) { c->error = "null image pointer"; return 1; }
		HIP_TRY(c, hipMemcpy2DAsync(dst + (size_t)i * stride + (size_t)kImgPad * c->pitch + kImgPad, (size_t)c->pitch * 4, src[i],
		                            (size_t)pitch_floats * 4, (size_t)c->W * 4, (size_t)c->H, kind, c->stream));
	}
	{   // border replication == clamp-to-edge addressing (APD.cpp:1511-1515)
		const long long cells = (long long)(2 * kImgPad * (c->W + 2 * kImgPad) + 2 * kImgPad * c->H) * c->NI;
		hipLaunchKernelGGL(dvp_pad_replicate, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, dst, c->W, c->H, c->pitch, stride, c->NI);
		HIP_TRY(c, hipGetLastError());
	}
	if (pairs) {
		const int PH = c->H + 2 * kImgPad;
		hipLaunchKernelGGL(dvp_interleave_rows, dim3((unsigned)((c->pitch + 255) / 256), (unsigned)PH, (unsigned)c->NI), dim3(256), 0, c->stream,
		                   dst, pairs, PH, c->pitch, stride, c->NI);
		HIP_TRY(c, hipGetLastError());
		// byte planes for 8-bit exact image sets (Dev::images8)
		HIP_TRY(c, hipMemsetAsync(c->images8_flag, 0, sizeof(int), c->stream));
		const int t8x = img8_tiles_x(c->W), t8y = img8_tiles_y(c->H);
		const size_t n = (size_t)t8x * t8y * 64 * c->NI;
		hipLaunchKernelGGL(dvp_pairs_to_tiles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, pairs, c->images8, c->W + 2 * kImgPad, PH, c->pitch, stride,
		                   t8x, t8y, c->NI, c->images8_flag);
		HIP_TRY(c, hipGetLastError());
		int inexact = 1;
		HIP_TRY(c, hipMemcpyAsync(&inexact, c->images8_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		c->images8_ok = inexact == 0 && !c->no_images8;
		sync_dev_struct(c);
	}
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}
int dvp_upload_images(dvp_ctx* c, const float* const* images, int pitch_floats) {
	c->anchor_tab_valid = false;
	return upload_planes(c, c->image_stage, images, pitch_floats, hipMemcpyHostToDevice, c->images);
}
int dvp_upload_images_device(dvp_ctx* c, const float* const* images, int pitch_floats) {
	c->anchor_tab_valid = false;
	return upload_planes(c, c->image_stage, images, pitch_floats, hipMemcpyDeviceToDevice, c->images);
}
static int ensure_depths(dvp_ctx* c) {
	if (!c->depths) {
		if (dalloc(c, &c->depths, (size_t)c->pitch * (c->H + 2 * kImgPad) * c->NI)) return 1;
		sync_dev_struct(c);
	}
	c->have_depths = true;
	return 0;
}
int dvp_upload_depths(dvp_ctx* c, const float* const* depths, int pitch_floats) {
	if (set_device(c) || ensure_depths(c)) return 1;
	return upload_planes(c, c->depths, depths, pitch_floats, hipMemcpyHostToDevice);
}
int dvp_upload_depths_device(dvp_ctx* c, const float* const* depths, int pitch_floats) {
	if (set_device(c) || ensure_depths(c)) return 1;
	return upload_planes(c, c->depths, depths, pitch_floats, hipMemcpyDeviceToDevice);
}

int dvp_upload_cameras(dvp_ctx* c, const DvpCamera* cams, int n) {
	if (set_device(c)) return 1;
	if (n != c->NI) { c->error = "dvp_upload_cameras: n != num_images"; return 1; }
	HIP_TRY(c, hipMemcpyAsync(c->cameras, cams, sizeof(DvpCamera) * n, hipMemcpyHostToDevice, c->stream));
	hipLaunchKernelGGL(dvp_prepare_views, dim3(1), dim3(64), 0, c->stream, c->cameras, c->views, n);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}

static int ensure_weak_buffers(dvp_ctx* c, size_t weak_count) {
	const size_t need = weak_count ? weak_count : 1;
	if (need > c->weak_alloc) {
		// The host driver recycles one context over all views and passes (APD.cpp pool): grow geometrically and give the
		// superseded blocks back.
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		if (c->side) HIP_TRY(c, hipStreamSynchronize(c->side));
		dfree(c, &c->neighbours); dfree(c, &c->gn_points); dfree(c, &c->gn_count); dfree(c, &c->complex_); dfree(c, &c->label_boundary);
		const size_t cap = std::min<size_t>(c->L, std::max(need, c->weak_alloc + c->weak_alloc / 2));
		if (dalloc(c, &c->neighbours, cap * DVP_NEIGHBOUR_NUM, false)) return 1;
		if (dalloc(c, &c->gn_points, cap * kGnDirSlots, false) || dalloc(c, &c->gn_count, cap)) return 1;
		if (dalloc(c, &c->complex_, cap)) return 1;
		if (dalloc(c, &c->label_boundary, cap * 8, false)) return 1;
		c->weak_alloc = cap;
	}
	HIP_TRY(c, hipMemsetAsync(c->neighbours, 0xFF, need * DVP_NEIGHBOUR_NUM * sizeof(s2), c->stream));   // (-1,-1)
	HIP_TRY(c, hipMemsetAsync(c->label_boundary, 0xFF, need * 8 * sizeof(s2), c->stream));
	HIP_TRY(c, hipMemsetAsync(c->complex_, 0, need * 4, c->stream));
	return 0;
}

// weak_info (on the device) -> neighbours_map, the compacted WEAK lists, the per-WEAK buffers
static int rebuild_weak_lists(dvp_ctx* c) {
	const size_t L = c->L;
	// weak_info -> neighbours_map (running index of WEAK pixels, APD.cpp:1182-1193) and the compacted WEAK lists of the
	// list kernels (black, then red; rows the reference's half grid never reaches, APD.cu:4421-4424, are kept like in the
	// full-grid launch), all on the device: see dvp_weak_counts / dvp_scan3 / dvp_weak_fill.  List order = 64 x 64
	// super-tiles, 16 x 16 tiles inside them, rows inside those: the lanes of a wave are neighbours in both directions and
	// the ~8 consecutive workgroups of a super-tile share their anchors and the source lines those touch; stage_body_list
	// hands such runs to ONE XCD.  (16x16 vs 16x8 / 32x4 / row-major: 592.7 / 599 / 609 / 622 ms per REFINE pass, r01.)
	// Every list kernel is order-independent (a WEAK pixel only reads STRONG pixels' state).
	const int supers_x = (c->W + kWeakSuper - 1) / kWeakSuper, supers_y = (c->H + kWeakSuper - 1) / kWeakSuper;
	const int n_slots = supers_x * supers_y * 16, n_chunks = (int)((L + kWeakChunk - 1) / kWeakChunk);
	const int n_units = n_slots + n_chunks;
	if (!c->weak_counts) {
		if (dalloc(c, &c->weak_counts, (size_t)2 * n_slots + n_chunks + 4)) return 1;
		HIP_TRY(c, hipHostMalloc((void**)&c->weak_totals_host, 4 * sizeof(int)));
	}
	int* totals = c->weak_counts + 2 * (size_t)n_slots + n_chunks;
	hipLaunchKernelGGL(dvp_weak_counts, dim3((n_units + 3) / 4), dim3(256), 0, c->stream, c->weak_info, c->W, c->H, supers_x, n_slots, n_chunks, c->weak_counts);
	hipLaunchKernelGGL(dvp_scan3, dim3(3), dim3(1024), 0, c->stream, c->weak_counts, 0, n_slots, 2 * n_slots, 2 * n_slots + n_chunks, totals);
	HIP_TRY(c, hipMemcpyAsync(c->weak_totals_host, totals, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	const int nb = c->weak_totals_host[0], nr = c->weak_totals_host[1], wc = c->weak_totals_host[2];
	if (nb + nr != wc) { c->error = "dvp_upload_state: WEAK list / map counts disagree"; return 1; }
	c->d.weak_count = wc;
	c->d.weak_black = nb;
	c->d.weak_red = nr;
	if ((size_t)wc > c->weak_list_alloc) {
		dfree(c, &c->weak_list);
		const size_t cap = std::min<size_t>(L, std::max<size_t>((size_t)wc, c->weak_list_alloc + c->weak_list_alloc / 2));
		if (dalloc(c, &c->weak_list, cap, false)) return 1;
		c->weak_list_alloc = cap;
	}
	if (wc > 0) {
		hipLaunchKernelGGL(dvp_weak_fill, dim3((n_units + 3) / 4), dim3(256), 0, c->stream, c->weak_info, c->W, c->H, supers_x, n_slots, n_chunks,
		                   c->weak_counts, totals, c->weak_list, c->neighbours_map);
		HIP_TRY(c, hipGetLastError());
	} else {
		HIP_TRY(c, hipMemsetAsync(c->neighbours_map, 0, L * 4, c->stream));
	}
	if (ensure_weak_buffers(c, (size_t)wc)) return 1;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	sync_dev_struct(c);
	return 0;
}

int dvp_upload_state(dvp_ctx* c, const float* planes, const uint32_t* views, const uint8_t* weak,
                     const uint8_t* edge, const int32_t* label, const int32_t* radius) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	const size_t L = c->L;
	if (planes) HIP_TRY(c, hipMemcpyAsync(c->planes, planes, L * 16, hipMemcpyHostToDevice, c->stream));
	if (views) HIP_TRY(c, hipMemcpyAsync(c->selected_views, views, L * 4, hipMemcpyHostToDevice, c->stream));
	if (edge) HIP_TRY(c, hipMemcpyAsync(c->edge, edge, L, hipMemcpyHostToDevice, c->stream));
	if (label) HIP_TRY(c, hipMemcpyAsync(c->label, label, L * 4, hipMemcpyHostToDevice, c->stream));
	if (radius) HIP_TRY(c, hipMemcpyAsync(c->radius, radius, L * 4, hipMemcpyHostToDevice, c->stream));
	if (weak) HIP_TRY(c, hipMemcpyAsync(c->weak_info, weak, L, hipMemcpyHostToDevice, c->stream));
	return rebuild_weak_lists(c);
}

// The per-pixel input state of a REFINE_INIT pass from the coarser level's result maps, up-sampled on the device
// (dvp_rescale_state): 25 bytes per COARSE pixel cross the bus instead of 25 per fine one, and the host neither rescales
// nor assembles planes.
int dvp_upload_state_rescaled(dvp_ctx* c, int src_w, int src_h, const float* depth, const float* normal_xyz, const uint32_t* views,
                              const uint8_t* weak, const int32_t* radius, int radius_fallback, const uint8_t* edge, const int32_t* label) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	if (src_w <= 0 || src_h <= 0 || !depth || !normal_xyz || !views) { c->error = "dvp_upload_state_rescaled: depth, normal and selected_views are required"; return 1; }
	const size_t L = c->L, n = (size_t)src_w * src_h;
	const size_t off_normal = n * 4, off_views = off_normal + n * 12, off_radius = off_views + n * 4, off_weak = off_radius + n * 4, bytes = off_weak + n;
	if (bytes > c->coarse_alloc) {
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		dfree(c, &c->coarse);
		if (dalloc(c, &c->coarse, bytes, false)) return 1;
		c->coarse_alloc = bytes;
	}
	HIP_TRY(c, hipMemcpyAsync(c->coarse, depth, n * 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->coarse + off_normal, normal_xyz, n * 12, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->coarse + off_views, views, n * 4, hipMemcpyHostToDevice, c->stream));
	if (radius) HIP_TRY(c, hipMemcpyAsync(c->coarse + off_radius, radius, n * 4, hipMemcpyHostToDevice, c->stream));
	if (weak) HIP_TRY(c, hipMemcpyAsync(c->coarse + off_weak, weak, n, hipMemcpyHostToDevice, c->stream));
	if (edge) HIP_TRY(c, hipMemcpyAsync(c->edge, edge, L, hipMemcpyHostToDevice, c->stream));
	if (label) HIP_TRY(c, hipMemcpyAsync(c->label, label, L * 4, hipMemcpyHostToDevice, c->stream));
	RescaleArgs a;
	a.W = c->W; a.H = c->H; a.sw = src_w; a.sh = src_h;
	a.scale_x = c->W / static_cast<float>(src_w);
	a.scale_y = c->H / static_cast<float>(src_h);
	a.depth = reinterpret_cast<const float*>(c->coarse); a.normal = reinterpret_cast<const float*>(c->coarse + off_normal);
	a.views = reinterpret_cast<const uint32_t*>(c->coarse + off_views);
	a.radius = radius ? reinterpret_cast<const int*>(c->coarse + off_radius) : nullptr;
	a.weak = weak ? c->coarse + off_weak : nullptr;
	a.planes = c->planes; a.out_views = c->selected_views; a.out_weak = c->weak_info; a.out_radius = c->radius;
	a.radius_fallback = radius_fallback;
	hipLaunchKernelGGL(dvp_rescale_state, dim3((unsigned)((c->W + 255) / 256), (unsigned)c->H), dim3(256), 0, c->stream, a);
	HIP_TRY(c, hipGetLastError());
	return rebuild_weak_lists(c);
}

int dvp_reset_state(dvp_ctx* c) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	const size_t L = c->L;
	HIP_TRY(c, hipMemsetAsync(c->planes, 0, L * 16, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->fit_planes, 0, L * 16, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->costs, 0, L * 4, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->selected_views, 0, (L + c->W) * 4, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->view_weight, 0, L * 32, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->weak_info, DVP_STRONG, L, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->edge, 0, L, c->stream));        // a recycled context must not see the previous view's priors
	HIP_TRY(c, hipMemsetAsync(c->label, 0, L * 4, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->neighbours_map, 0, L * 4, c->stream));
	HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->radius, c->d.params.strong_radius, L, c->stream));
	c->d.weak_count = 0;
	c->d.weak_black = c->d.weak_red = 0;
	if (ensure_weak_buffers(c, 0)) return 1;
	sync_dev_struct(c);
	return 0;
}

// Device-side snapshot of the per-pixel INPUT state of a pass (what dvp_upload_state delivers:
// planes, selected_views, weak_info, radius; the WEAK-pixel list and neighbours_map derived from
// weak_info stay valid because the same weak map comes back) and its restore, which also returns the
// pass-internal buffers to their freshly-uploaded content.  Lets a caller run the same pass again
// from identical inputs (bench steps, A/B checks) without a host round trip.
int dvp_image_format(const dvp_ctx* c) { return (c && c->images8_ok) ? 1 : 0; }
int dvp_save_state(dvp_ctx* c) {
	if (set_device(c)) return 1;
	const size_t L = c->L;
	if (!c->saved_planes) {
		if (dalloc(c, &c->saved_planes, L, false) || dalloc(c, &c->saved_views, L, false) ||
			dalloc(c, &c->saved_weak, L, false) || dalloc(c, &c->saved_radius, L, false)) return 1;
	}
	HIP_TRY(c, hipMemcpyAsync(c->saved_planes, c->planes, L * 16, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->saved_views, c->selected_views, L * 4, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->saved_weak, c->weak_info, L, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->saved_radius, c->radius, L * 4, hipMemcpyDeviceToDevice, c->stream));
	c->have_saved = true;
	return 0;
}
int dvp_restore_state(dvp_ctx* c) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	if (!c->have_saved) { c->error = "dvp_restore_state: no saved state"; return 1; }
	const size_t L = c->L;
	HIP_TRY(c, hipMemcpyAsync(c->planes, c->saved_planes, L * 16, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->selected_views, c->saved_views, L * 4, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->weak_info, c->saved_weak, L, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->radius, c->saved_radius, L * 4, hipMemcpyDeviceToDevice, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->fit_planes, 0, L * 16, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->costs, 0, L * 4, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->view_weight, 0, L * 32, c->stream));
	HIP_TRY(c, hipMemsetAsync(c->weak_reliable, 0, L, c->stream));
	if (ensure_weak_buffers(c, (size_t)c->d.weak_count)) return 1;
	return 0;
}

int dvp_set_params(dvp_ctx* c, const DvpParams* p) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	if (p->num_images != c->NI) { c->error = "dvp_set_params: params.num_images != context num_images"; return 1; }
	if (p->use_edge == 0) { c->error = "dvp_set_params: use_edge=false is rejected: the reference's legacy ACMH branch (APD.cu:2142-2460) adopts planes through positions[], which only the use_edge branch assigns (APD.cu:2036, 2084, 2133 vs 2559-2563) - it has no defined result to reproduce"; return 1; }
	if (p->weak_radius < 0 || p->weak_radius > 15) { c->error = "dvp_set_params: weak_radius out of range [0,15]"; return 1; }
	c->d.params = *p;
	set_neighbour_consts(&c->d);
	if (c->lut_radius != p->weak_radius) {
		std::vector<int> taps, start;
		make_sector_taps(p->weak_radius, &taps, &start);
		if (dalloc(c, &c->sector_taps, taps.size(), false) || dalloc(c, &c->sector_start, start.size(), false)) return 1;
		HIP_TRY(c, hipMemcpyAsync(c->sector_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->sector_start, start.data(), start.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		c->lut_radius = p->weak_radius;
	}
	sync_dev_struct(c);
	return 0;
}
int dvp_set_seed(dvp_ctx* c, uint64_t seed) { c->d.seed = seed; return 0; }
int dvp_set_sampler(dvp_ctx* c, int s) { c->d.sampler = s ? 1 : 0; return 0; }
int dvp_set_profiling(dvp_ctx* c, int on) { c->profiling = on != 0; sync_dev_struct(c); return 0; }

// ---- launches ---------------------------------------------------------------------------------
// `fused` (dvp_run_patchmatch only): DepthToWeak does LocalRefine too; GenEdgeInform leaves the visibility-prior
// candidates to dvp_run_patchmatch (side stream; not computed at all when the pass has no WEAK pixel: their only
// reader is the weak update's anchor_cost)
// The pass' table of anchor reference sides (dvp_weak_wave.hpp: build_anchor_record) — built at the first weak update after
// anything that can change the anchors (GenNeighbours, NeigbourUpdate), the offsets (GenEdgeInform), the images or the WEAK
// list; the iterations of a pass then share it.  A table that does not fit is not an error: the kernels that form the
// reference side per item give the same bits.
// room in the anchor table for `wc` WEAK pixels (grown with half as much again on top: the views of a level differ in their WEAK
// counts, and every regrow is a device-wide free + a multi-GB allocation)
static int grow_anchor_table(dvp_ctx* c, size_t wc) {
	const size_t need = wc * (size_t)(c->NI - 1) * kAnchors;
	if (c->anchor_tab_off || need <= c->anchor_tab_alloc) return 0;
	{
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		dfree(c, &c->anchor_tab);
		c->anchor_tab_alloc = 0;
		const size_t cap = need + need / 2;
		void* q = nullptr;
		size_t got = cap;
		if (hipMalloc(&q, cap * sizeof(AnchorRec)) != hipSuccess) {
			(void)hipGetLastError();
			got = need;
			if (hipMalloc(&q, need * sizeof(AnchorRec)) != hipSuccess) q = nullptr;
		}
		if (!q) {
			(void)hipGetLastError();
			c->anchor_tab_off = true;
			sync_dev_struct(c);
			fprintf(stderr, "dvp: no room for the weak update's anchor table (%.1f GB); forming the reference side per item\n", (double)need * sizeof(AnchorRec) / 1e9);
			return 0;
		}
		c->allocs.push_back(q);
		c->anchor_tab = (AnchorRec*)q;
		c->anchor_tab_alloc = got;
		sync_dev_struct(c);
	}
	return 0;
}
static int ensure_anchor_table(dvp_ctx* c, int covered_rows) {
	if (c->anchor_tab_off) return 0;
	if (c->anchor_tab_valid) return 0;
	const int wc = c->d.weak_black + c->d.weak_red;
	const size_t need = (size_t)wc * (size_t)(c->NI - 1) * kAnchors;
	if (grow_anchor_table(c, (size_t)wc)) return 1;
	if (c->anchor_tab_off) return 0;
	c->anchor_tab_valid = true;
	sync_dev_struct(c);
	ListArgs la;
	la.base = 0; la.count = wc; la.iter = 0; la.covered_rows = covered_rows;
	const long long n = (long long)need;
	if (n > 0) {
		hipLaunchKernelGGL(dvp_weak_anchor_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d, la);
		HIP_TRY(c, hipGetLastError());
	}
	return 0;
}

// hand-over buffers of the phased weak update (624 + 32 S bytes per WEAK pixel: 1.6 GB at 6208x4128 with 7 % WEAK, S = 9).  A
// context that cannot have them keeps the one-wave kernel: the same bits.
static int grow_weak_phase_buffers(dvp_ctx* c, size_t wc);
static int ensure_weak_phase_buffers(dvp_ctx* c) { return grow_weak_phase_buffers(c, (size_t)(c->d.weak_black + c->d.weak_red)); }
static int grow_weak_phase_buffers(dvp_ctx* c, size_t wc) {
	if (!c->weak_phased) return 0;
	if (wc <= c->weak_phase_alloc) return 0;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	dfree(c, &c->weak_rec); dfree(c, &c->weak_ctab); dfree(c, &c->weak_ev);
	c->weak_phase_alloc = 0;
	const size_t cap = std::min<size_t>(c->L, wc + wc / 2), S = (size_t)c->NI - 1;
	void *r = nullptr, *t = nullptr, *e = nullptr;
	if (getenv("DVP_TEST_WEAK_PHASE_ALLOC_FAIL") /* test hook: take the fallback */ || hipMalloc(&r, cap * sizeof(WeakRec)) != hipSuccess ||
	    hipMalloc(&t, cap * kTaps * kTaps * sizeof(f2)) != hipSuccess || hipMalloc(&e, cap * 8 * S * sizeof(float)) != hipSuccess) {
		(void)hipGetLastError();
		if (r) (void)hipFree(r);
		if (t) (void)hipFree(t);
		c->weak_phased = false;
		fprintf(stderr, "dvp: no room for the phased weak update's hand-over buffers (%.2f GB); using the one-wave kernel\n", (double)cap * (sizeof(WeakRec) + 288 + 32 * S) / 1e9);
		sync_dev_struct(c);
		return 0;
	}
	c->allocs.push_back(r); c->allocs.push_back(t); c->allocs.push_back(e);
	c->weak_rec = (WeakRec*)r; c->weak_ctab = (f2*)t; c->weak_ev = (float*)e;
	c->weak_phase_alloc = cap;
	sync_dev_struct(c);
	return 0;
}

// The optional buffers of the split strong update and of the view-compacted sweep passes: allocated at the first launch that
// wants them — or ahead of it by dvp_ctx_reserve, off the critical path (a multi-GB hipMalloc inside a launch site took 0.5-1 s
// of a view on some boxes: profiles/r06_ab_notes.txt).  A context that cannot have them keeps the monolithic / fused kernels.
static void ensure_strong_split_buffers(dvp_ctx* c) {
	if (c->slot_costs || !c->strong_split || c->NI - 1 > 16) return;
	// 17 x S floats per pixel of one colour (7.8 GB at 6208x4128, S = 9): a part that cannot spare them runs the
	// monolithic kernel instead, which gives the same bits (test_strong_update_forms_equal_the_oracle)
	const size_t Lh = (size_t)((c->W + 1) / 2) * c->H;
	void *sc = nullptr, *sr = nullptr;
	if (getenv("DVP_TEST_SPLIT_ALLOC_FAIL") /* test hook: take the fallback */ || hipMalloc(&sc, (size_t)kSlotCount * (c->NI - 1) * Lh * sizeof(*c->slot_costs) + 64 /* load_slot_costs reads whole 16-byte pieces */) != hipSuccess || hipMalloc(&sr, (size_t)SR_FIELDS * Lh * sizeof(*c->strong_rec)) != hipSuccess) {
		(void)hipGetLastError();   // clear the sticky out-of-memory status
		if (sc) (void)hipFree(sc);
		c->strong_split = false;
		fprintf(stderr, "dvp: no room for the split strong update's cost buffer (%.1f GB); using the monolithic kernel\n", (double)kSlotCount * (c->NI - 1) * Lh * 4 / 1e9);
	} else {
		c->allocs.push_back(sc); c->allocs.push_back(sr);
		c->slot_costs = (decltype(c->slot_costs))sc; c->strong_rec = (decltype(c->strong_rec))sr;
		sync_dev_struct(c);
	}
}
static void ensure_sweep_buffers(dvp_ctx* c) {
	if (c->sweep_cost || !c->sweep_split) return;
	// 73 floats per (pixel, view) + 61 + 8 per pixel (67 GB at 6208x4128, S = 9): a context that cannot have them keeps the fused kernel
	const size_t L = c->L;
	void *r = nullptr, *sc = nullptr, *pc = nullptr;
	// the cost records of the whole image, or of a band of rows when the context is told to keep them small (fresh device memory is 31-40 ms
	// per GB on this part, tools/micro/alloc_time.hip: 2.3 s for a 25-Mpx view's 67 GB)
	size_t band_px = L;
	c->sweep_band_rows = 0;
	if (c->sweep_band_gb > 0.0) {
		const double whole = (double)sweep_cost_floats(L, c->NI - 1) * sizeof(float);
		const int bands = (int)std::ceil(whole / (c->sweep_band_gb * 1e9));
		if (bands > 1) {
			int rows = (c->H + bands - 1) / bands;
			rows = (rows + kSweepRows - 1) / kSweepRows * kSweepRows;
			if (rows < c->H) { c->sweep_band_rows = rows; band_px = (size_t)rows * c->W; }
		}
	}
	if (getenv("DVP_TEST_SWEEP_ALLOC_FAIL") || hipMalloc(&r, 2 * L * sizeof(f4)) != hipSuccess || hipMalloc(&pc, 61 * L * sizeof(float)) != hipSuccess ||
	    hipMalloc(&sc, sweep_cost_floats(band_px, c->NI - 1) * sizeof(float)) != hipSuccess) {
		(void)hipGetLastError();
		if (r) (void)hipFree(r);
		if (pc) (void)hipFree(pc);
		c->sweep_split = false;
		fprintf(stderr, "dvp: no room for the view-compacted DepthToWeak's cost buffer (%.1f GB); using the fused kernel\n", (double)(c->NI - 1) * kSweepFields * L * 4 / 1e9);
	} else {
		c->allocs.push_back(r); c->allocs.push_back(sc); c->allocs.push_back(pc);
		c->sweep_rec = (f4*)r; c->sweep_cost = (float*)sc; c->sweep_pc = (float*)pc;
		sync_dev_struct(c);
	}
}

static int launch_stage(dvp_ctx* c, int stage, int iter, int colour, bool fused = false) {
	if (stage < 0 || stage >= DVP_ST_LAUNCHABLE) { c->error = "bad stage id"; return 1; }
	if (!c->sector_taps) { c->error = "dvp_set_params must be called before running kernels"; return 1; }
	if (c->d.params.geom_consistency && !c->have_depths) { c->error = "geom_consistency is on but no depth maps were uploaded"; return 1; }
	const LaunchGeom g = make_geom(c->W, c->H, stage_is_half(stage));
	LaunchArgs a;
	a.tiles_x = g.tiles_x; a.tiles = g.tiles; a.rows = g.rows; a.half = g.half ? 1 : 0;
	a.colour = colour; a.iter = iter;
	if (c->events.size() >= 2048 && dvp_get_timings(c, nullptr)) return 1;   // fold pending timings: bounds the event pool
	if (stage != DVP_ST_STRONG_UPDATE && stage != DVP_ST_RANSAC_FIT && stage != DVP_ST_WEAK_UPDATE) c->anchor_tab_valid = false;   // anchors / offsets / WEAK states may change: the next weak update rebuilds its table
	if (stage == DVP_ST_STRONG_UPDATE) {
		// pre-launch snapshot: the direction-4 samples of the strong update are same-colour pixels
		// (APD.cu:2071-2074); every neighbour read of that kernel sees the state before the launch.
		// Then the light sample-search launch.  Both are timed in their own bucket (DVP_ST_STRONG_PREP)
		// so that stage_ms[DVP_ST_STRONG_UPDATE] is the update kernel alone.
		EventPair prep;
		prep.stage = DVP_ST_STRONG_PREP;
		HIP_TRY(c, hipEventCreate(&prep.a));
		HIP_TRY(c, hipEventCreate(&prep.b));
		HIP_TRY(c, hipEventRecord(prep.a, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->planes_snap, c->planes, c->L * 16, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->costs_snap, c->costs, c->L * 4, hipMemcpyDeviceToDevice, c->stream));
		hipLaunchKernelGGL(dvp_strong_search, dim3(g.grid()), dim3(256), 0, c->stream, c->d, a);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipEventRecord(prep.b, c->stream));
		c->events.push_back(prep);
	}
	EventPair ep;
	ep.stage = stage;
	HIP_TRY(c, hipEventCreate(&ep.a));
	HIP_TRY(c, hipEventCreate(&ep.b));
	if (c->profiling) HIP_TRY(c, hipMemsetAsync(c->eval_counter, 0, 8, c->stream));
	HIP_TRY(c, hipEventRecord(ep.a, c->stream));
	const dim3 grid(g.grid()), block(256);
	const dim3 wave_grid((unsigned)((g.grid() + 7) / 8 * 32)), wave_block(64);   // DVP_KERNEL64 launch sites: four one-wave workgroups per tile
	const bool list_stage = stage == DVP_ST_FIND_NEAREST_STRONG || stage == DVP_ST_GEN_NEIGHBOURS || stage == DVP_ST_NEIGHBOUR_UPDATE ||
	                        stage == DVP_ST_RANSAC_FIT || stage == DVP_ST_WEAK_UPDATE;
	if (stage == DVP_ST_FIND_NEAREST_STRONG && c->d.weak_black + c->d.weak_red > 0) {   // its ring search reads row and column segments of the STRONG map
		const size_t words = edge_bits_words(c->W, c->H);
		hipLaunchKernelGGL(dvp_pack_edge_bits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, c->weak_info, c->strong_bits, c->W, c->H, edge_tiles_x(c->W), words, (int)DVP_STRONG);
		hipLaunchKernelGGL(dvp_pack_bits_transposed, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, c->weak_info, c->strong_bits_t, c->W, c->H, edge_tiles_x(c->W), words, (int)DVP_STRONG);
		HIP_TRY(c, hipGetLastError());
	}
	if (stage == DVP_ST_GEN_NEIGHBOURS || stage == DVP_ST_RANSAC_FIT) {   // the launch sites that walk lines over the edge map
		const size_t words = edge_bits_words(c->W, c->H);
		hipLaunchKernelGGL(dvp_pack_edge_bits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, c->edge, c->edge_bits, c->W, c->H, edge_tiles_x(c->W), words, -1);
		HIP_TRY(c, hipGetLastError());
		const int cells = c->d.sat_cells_x * c->d.sat_cells_y;
		hipLaunchKernelGGL(dvp_edge_cell_counts, dim3((cells + 255) / 256), dim3(256), 0, c->stream, c->d);
		hipLaunchKernelGGL(dvp_edge_sat_rows, dim3((c->d.sat_cells_y + 1 + 63) / 64), dim3(64), 0, c->stream, c->d);
		hipLaunchKernelGGL(dvp_edge_sat_cols, dim3((c->d.sat_cells_x + 1 + 63) / 64), dim3(64), 0, c->stream, c->d);
		HIP_TRY(c, hipGetLastError());
		if (stage == DVP_ST_GEN_NEIGHBOURS) {   // its anchor search probes "is this pixel STRONG" all over the image
			hipLaunchKernelGGL(dvp_pack_edge_bits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream, c->weak_info, c->strong_bits, c->W, c->H, edge_tiles_x(c->W), words, (int)DVP_STRONG);
			HIP_TRY(c, hipGetLastError());
		}
	}
	if (list_stage) {
		// weak-path launch sites: lane-per-WEAK-pixel.  Non-WEAK pixels' outputs of these kernels are
		// constants / copies and are produced by plain fills: weak_nearest_strong = (-1,-1)
		// (APD.cu:4169-4173), fit plane = plane (APD.cu:4208-4211).
		if (stage == DVP_ST_FIND_NEAREST_STRONG) HIP_TRY(c, hipMemsetAsync(c->weak_nearest_strong, 0xFF, c->L * sizeof(s2), c->stream));
		if (stage == DVP_ST_RANSAC_FIT) HIP_TRY(c, hipMemcpyAsync(c->fit_planes, c->planes, c->L * 16, hipMemcpyDeviceToDevice, c->stream));
		ListArgs la;
		la.iter = iter;
		la.covered_rows = 2 * make_geom(c->W, c->H, true).rows;
		la.base = (stage == DVP_ST_WEAK_UPDATE && colour == 1) ? c->d.weak_black : 0;
		la.count = (stage == DVP_ST_WEAK_UPDATE) ? (colour == 0 ? c->d.weak_black : (colour == 1 ? c->d.weak_red : c->d.weak_black + c->d.weak_red)) : c->d.weak_black + c->d.weak_red;   // colour 2 (dvp_run_patchmatch): both colours of the weak update in one launch site
		if (la.count > 0) {
			const dim3 lg((la.count + 255) / 256);
			const bool ex = c->d.sampler != 0;
			switch (stage) {
			case DVP_ST_FIND_NEAREST_STRONG: hipLaunchKernelGGL(ex ? dvp_find_nearest_strong_list_exact : dvp_find_nearest_strong_list, lg, block, 0, c->stream, c->d, la); break;
			case DVP_ST_GEN_NEIGHBOURS:
				if (c->gn_wave) hipLaunchKernelGGL(dvp_gen_neighbours_search, dim3((la.count + 3) / 4), block, 0, c->stream, c->d, la);
				else hipLaunchKernelGGL(dvp_gen_neighbours_list, lg, block, 0, c->stream, c->d, la);
				hipLaunchKernelGGL(dvp_gen_neighbours_fit, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
				break;
			case DVP_ST_NEIGHBOUR_UPDATE: hipLaunchKernelGGL(ex ? dvp_neighbour_update_list_exact : dvp_neighbour_update_list, lg, block, 0, c->stream, c->d, la); break;
			case DVP_ST_RANSAC_FIT:
				if (c->ransac_wave) hipLaunchKernelGGL(dvp_ransac_fit_plane_wave, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
				else hipLaunchKernelGGL(ex ? dvp_ransac_fit_plane_list_exact : dvp_ransac_fit_plane_list, lg, block, 0, c->stream, c->d, la);
				break;
			case DVP_ST_WEAK_UPDATE:
				if (ensure_anchor_table(c, la.covered_rows)) return 1;
				if (c->d.anchor_tab && ensure_weak_phase_buffers(c)) return 1;
				// (a few thousand WEAK pixels do not fill the machine in any form: the eight launches then cost more than they save —
				// 4.4 against 1.7 ms for the weak updates of a 3104x2064 view with 0.2 % WEAK pixels — and the one-wave kernel takes them)
				if (c->d.anchor_tab && c->weak_phased && (la.count >= c->weak_phased_min || c->weak_phased_min <= 0)) {
					const bool u8 = c->images8_ok;
					const dim3 w64(64), lg64((la.count + 63) / 64);
#define DVP_PICK(NAME) (ex ? (u8 ? NAME##_exact_u8 : NAME##_exact) : (u8 ? NAME##_u8 : NAME))
#define DVP_GROUP_LAUNCH(NAME, PHASE) { la.group = std::min(c->weak_group[PHASE], PHASE == 0 ? kGrpWide : kGrp); la.run = c->weak_run[PHASE]; hipLaunchKernelGGL(DVP_PICK(NAME), dim3((la.count + la.group - 1) / la.group), w64, 0, c->stream, c->d, la); }
					DVP_GROUP_LAUNCH(dvp_weak_eval_candidates, 0)
					hipLaunchKernelGGL(dvp_weak_select_views, lg, block, 0, c->stream, c->d, la);
					DVP_GROUP_LAUNCH(dvp_weak_eval_planes, 1)
					hipLaunchKernelGGL(dvp_weak_make_hypotheses, lg, block, 0, c->stream, c->d, la);
					DVP_GROUP_LAUNCH(dvp_weak_eval_first_view, 2)
					DVP_GROUP_LAUNCH(dvp_weak_eval_survivors, 3)
#undef DVP_GROUP_LAUNCH
					hipLaunchKernelGGL(dvp_weak_adopt, lg, block, 0, c->stream, c->d, la);
					hipLaunchKernelGGL(ex ? dvp_weak_final_cost_exact : dvp_weak_final_cost, lg64, w64, 0, c->stream, c->d, la);
#undef DVP_PICK
				} else if (c->d.anchor_tab) {
					if (c->images8_ok) hipLaunchKernelGGL(ex ? dvp_weak_update_wave_exact_u8 : dvp_weak_update_wave_u8, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
					else hipLaunchKernelGGL(ex ? dvp_weak_update_wave_exact : dvp_weak_update_wave, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
				} else {
					if (c->images8_ok) hipLaunchKernelGGL(ex ? dvp_weak_update_wave_exact_u8_notab : dvp_weak_update_wave_u8_notab, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
					else hipLaunchKernelGGL(ex ? dvp_weak_update_wave_exact_notab : dvp_weak_update_wave_notab, dim3(la.count), dim3(64), 0, c->stream, c->d, la);
				}
				break;
			}
			HIP_TRY(c, hipGetLastError());
		}
	} else {
	if (stage == DVP_ST_GEN_EDGE_INFORM && c->d.params.use_edge) {
		hipLaunchKernelGGL(dvp_edge_rays, dim3((c->W + c->H + 255) / 256, 8), dim3(256), 0, c->stream, c->d, 0);
		HIP_TRY(c, hipGetLastError());
	}
	if (stage == DVP_ST_GEN_EDGE_INFORM && c->d.params.use_label && c->d.weak_count > 0) {   // label boundaries are searched for WEAK pixels only
		if (!c->label_stop) {
			if (dalloc(c, &c->label_stop, c->L * 8, false)) return 1;
			sync_dev_struct(c);
		}
		hipLaunchKernelGGL(dvp_edge_rays, dim3((c->W + c->H + 255) / 256, 8), dim3(256), 0, c->stream, c->d, 1);
		HIP_TRY(c, hipGetLastError());
	}
	switch (stage) {
	case DVP_ST_GEN_EDGE_INFORM:
		if (!fused) {   // fused: dvp_run_patchmatch issued them on the side stream
			if (gen_candidates_all_views(c->d)) hipLaunchKernelGGL(dvp_gen_candidates_views, dim3(g.grid(), (unsigned)((c->NI - 1 + kCandGroup - 1) / kCandGroup)), block, 0, c->stream, c->d, a);
			else hipLaunchKernelGGL(dvp_gen_candidates, dim3(g.grid(), (unsigned)(c->NI - 1)), block, 0, c->stream, c->d, a);
		}
		hipLaunchKernelGGL(c->d.sampler ? dvp_gen_edge_inform_exact : dvp_gen_edge_inform, grid, block, 0, c->stream, c->d, a);
		break;
	case DVP_ST_RANDOM_INIT: hipLaunchKernelGGL(c->d.sampler ? dvp_random_init_exact : dvp_random_init, grid, block, 0, c->stream, c->d, a); break;
	case DVP_ST_STRONG_UPDATE:
		if (c->strong_split && c->NI - 1 <= 16) {
			ensure_strong_split_buffers(c);
		}
		if (c->strong_split && c->NI - 1 <= 16) {
			const int S = c->NI - 1;
			if (c->eval_items) hipLaunchKernelGGL(c->d.sampler ? dvp_strong_eval_items_exact : dvp_strong_eval_items, wave_grid, wave_block, 0, c->stream, c->d, a);
			else hipLaunchKernelGGL(c->d.sampler ? dvp_strong_eval_exact : dvp_strong_eval, wave_grid, wave_block, 0, c->stream, c->d, a);
			if (S <= 4) hipLaunchKernelGGL(dvp_strong_decide_v4, grid, block, 0, c->stream, c->d, a);
			else if (S <= 6) hipLaunchKernelGGL(dvp_strong_decide_v6, grid, block, 0, c->stream, c->d, a);
			else if (S <= 8) hipLaunchKernelGGL(dvp_strong_decide_v8, grid, block, 0, c->stream, c->d, a);
			else if (S <= 10) hipLaunchKernelGGL(dvp_strong_decide_v10, grid, block, 0, c->stream, c->d, a);
			else if (S <= 12) hipLaunchKernelGGL(dvp_strong_decide_v12, grid, block, 0, c->stream, c->d, a);
			else hipLaunchKernelGGL(dvp_strong_decide_v16, grid, block, 0, c->stream, c->d, a);
			// every lane on its own (hypothesis, view) sequence; the lanes' image planes are 32-bit byte offsets from the set's base
			if (c->refine_lanes && (size_t)c->pitch * (c->H + 2 * kImgPad) * 8 * c->NI < ((size_t)1 << 32))
				hipLaunchKernelGGL(c->d.sampler ? dvp_strong_refine_lanes_exact : dvp_strong_refine_lanes, wave_grid, wave_block, 0, c->stream, c->d, a);
			else hipLaunchKernelGGL(c->d.sampler ? dvp_strong_refine_exact : dvp_strong_refine, wave_grid, wave_block, 0, c->stream, c->d, a);
		}
		else if (c->NI - 1 <= kNarrowViews) hipLaunchKernelGGL(c->d.sampler ? dvp_strong_update_v8_exact : dvp_strong_update_v8, grid, block, 0, c->stream, c->d, a);
		else if (c->NI - 1 <= 16) hipLaunchKernelGGL(c->d.sampler ? dvp_strong_update_v16_exact : dvp_strong_update_v16, grid, block, 0, c->stream, c->d, a);
		else hipLaunchKernelGGL(c->d.sampler ? dvp_strong_update_exact : dvp_strong_update, grid, block, 0, c->stream, c->d, a);
		break;
	case DVP_ST_GET_DEPTH_NORMAL: hipLaunchKernelGGL(c->d.sampler ? dvp_get_depth_normal_exact : dvp_get_depth_normal, grid, block, 0, c->stream, c->d, a); break;
	case DVP_ST_FILTER_STRONG: hipLaunchKernelGGL(c->d.sampler ? dvp_filter_strong_exact : dvp_filter_strong, grid, block, 0, c->stream, c->d, a); break;
	case DVP_ST_DEPTH_TO_WEAK: {
		// the passes pay where the views diverge and the geometric term rides along (cfg3: 622 -> 536 ms, cfg5: 48 -> 44); a pass
		// without the geometric term (FIRST_INIT; cfg2, S = 5) measures 63 ms against the fused kernel's 59: DVP_SWEEP_SPLIT=2 forces the passes there too
		const bool sweep_passes = fused && c->sweep_split && (c->d.params.geom_consistency || c->sweep_force);
		if (sweep_passes) ensure_sweep_buffers(c);
		if (sweep_passes && c->sweep_split) {
			const bool ex = c->d.sampler != 0;
			LaunchArgs s0 = a, s1 = a, sb = a;
			s0.iter = 0; s1.iter = 1; sb.iter = kSweepBorderOnly;
			const int etx = (c->W + 63) / 64, ety = (c->H + kSweepRows - 1) / kSweepRows;
			s0.tiles_x = s1.tiles_x = etx; s0.tiles = s1.tiles = etx * ety;
			const dim3 egrid((unsigned)(etx * ety), (unsigned)(c->NI - 1));
			hipLaunchKernelGGL(dvp_sweep_prepare, grid, block, 0, c->stream, c->d, a);
			// evaluate / decide / evaluate / decide over the whole image, or band of rows after band of rows through the one band-sized
			// cost buffer (every pass is per pixel: same bits)
			const int band_rows = c->sweep_band_rows > 0 ? c->sweep_band_rows : c->H;
			for (int row0 = 0; row0 < c->H; row0 += band_rows) {
				Dev db = c->d;
				LaunchArgs b0 = s0, b1 = s1;
				dim3 bgrid = egrid;
				b0.rows = b1.rows = 0;
				if (c->sweep_band_rows > 0) {
					const int row1 = std::min(c->H, row0 + band_rows);
					db.sweep_px0 = row0 * c->W; db.sweep_row0 = row0; db.sweep_row1 = row1;
					const int bty = (row1 - row0 + kSweepRows - 1) / kSweepRows;
					b0.rows = b1.rows = row0 / kSweepRows;
					b0.tiles = b1.tiles = etx * bty;
					bgrid = dim3((unsigned)(etx * bty), (unsigned)(c->NI - 1));
				}
				hipLaunchKernelGGL(ex ? dvp_sweep_eval_exact : dvp_sweep_eval, bgrid, dim3(64), 0, c->stream, db, b0);
				hipLaunchKernelGGL(dvp_sweep_decide1, grid, block, 0, c->stream, db, a);
				if (sweep_window(c->d.params) < 30) {
					hipLaunchKernelGGL(ex ? dvp_sweep_eval_exact : dvp_sweep_eval, bgrid, dim3(64), 0, c->stream, db, b1);
				}
				hipLaunchKernelGGL(dvp_sweep_decide2, grid, block, 0, c->stream, db, a);
			}
			if (c->W >= 12 && c->H >= 12) {
				const long long frame = 12ll * c->W + 12ll * (c->H - 12);
				hipLaunchKernelGGL(ex ? dvp_sweep_border_exact : dvp_sweep_border, dim3((unsigned)((frame + 63) / 64)), dim3(64), 0, c->stream, c->d, sb);
			}
			else hipLaunchKernelGGL(ex ? dvp_depth_to_weak_refine_exact : dvp_depth_to_weak_refine, grid, block, 0, c->stream, c->d, sb);
		}
		else if (fused) hipLaunchKernelGGL(c->d.sampler ? dvp_depth_to_weak_refine_exact : dvp_depth_to_weak_refine, grid, block, 0, c->stream, c->d, a);
		else hipLaunchKernelGGL(c->d.sampler ? dvp_depth_to_weak_exact : dvp_depth_to_weak, grid, block, 0, c->stream, c->d, a);
		break;
	}
	case DVP_ST_LOCAL_REFINE: hipLaunchKernelGGL(c->d.sampler ? dvp_local_refine_exact : dvp_local_refine, grid, block, 0, c->stream, c->d, a); break;
	}
	HIP_TRY(c, hipGetLastError());
	}
	HIP_TRY(c, hipEventRecord(ep.b, c->stream));
	c->events.push_back(ep);
	if (c->profiling) {
		unsigned long long n = 0;
		HIP_TRY(c, hipMemcpyAsync(&n, c->eval_counter, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		c->timings.ncc_evals[stage] += n;
	}
	return 0;
}

int dvp_run_stage(dvp_ctx* c, int stage, int iter, int colour) {
	if (set_device(c)) return 1;
	return launch_stage(c, stage, iter, colour);
}

// Optional buffers ahead of their first use (a helper thread of the driver calls this on the context it prepares for the next
// pyramid level): flags bit 0 = the split strong update's cost block, bit 1 = the view-compacted sweep passes' buffers;
// weak_pixels > 0: anchor table and hand-over buffers of the weak update for that many WEAK pixels.  Nothing here is required —
// every launch site allocates what it lacks — and a buffer that does not fit selects the fall-back form exactly as there.
int dvp_ctx_reserve(dvp_ctx* c, int weak_pixels, int flags) {
	if (set_device(c)) return 1;
	if (flags & 1) ensure_strong_split_buffers(c);
	if (flags & 2) ensure_sweep_buffers(c);
	if (weak_pixels > 0) {
		const size_t wc = std::min<size_t>((size_t)weak_pixels, c->L);
		if (grow_anchor_table(c, wc)) return 1;
		if (c->anchor_tab && grow_weak_phase_buffers(c, wc)) return 1;
		c->anchor_tab_valid = false;
	}
	return 0;
}

int dvp_synchronize(dvp_ctx* c) {
	if (set_device(c)) return 1;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}

// APD::RunPatchMatch (APD.cu:4406-4532): same launch order; no host sync between launches.
int dvp_run_patchmatch(dvp_ctx* c) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	EventPair tot, itl;
	tot.stage = EV_TOTAL; itl.stage = EV_ITER_LOOP;
	HIP_TRY(c, hipEventCreate(&tot.a)); HIP_TRY(c, hipEventCreate(&tot.b));
	HIP_TRY(c, hipEventCreate(&itl.a)); HIP_TRY(c, hipEventCreate(&itl.b));
	HIP_TRY(c, hipEventRecord(tot.a, c->stream));
	// The candidates read the images, the sector tables and selected_views (which RandomInitialization is the first
	// to write) and are read by the weak updates: they run on the side stream from here to just before RandomInit.
	const bool side_work = c->d.weak_count > 0;
	bool masked = side_work && c->cand_mask_on && (c->cand_mask_mode == 1 || (c->cand_mask_mode < 0 && (size_t)c->d.weak_count * 25 < c->L));
	if (masked && !c->cand_mask) {   // mark bytes, anchor list, snapshot of the selected-view map: 9 bytes per pixel
		void *m = nullptr, *l = nullptr, *n = nullptr, *sv = nullptr;
		if (hipMalloc(&m, c->L) != hipSuccess || hipMalloc(&l, c->L * 4) != hipSuccess || hipMalloc(&n, 4) != hipSuccess || hipMalloc(&sv, (c->L + c->W) * 4) != hipSuccess) {
			(void)hipGetLastError();
			for (void* p : { m, l, n, sv }) if (p) (void)hipFree(p);
			c->cand_mask_on = masked = false;   // no room: every pixel, as before
		} else {
			for (void* p : { m, l, n, sv }) c->allocs.push_back(p);
			c->cand_mask = (uint8_t*)m; c->cand_list = (unsigned*)l; c->cand_n = (unsigned*)n; c->sel_snap = (uint32_t*)sv;
		}
	}
	if (side_work && !masked) {
		if (!c->sector_taps) { c->error = "dvp_set_params must be called before running kernels"; return 1; }
		HIP_TRY(c, hipEventRecord(c->side_fork, c->stream));
		HIP_TRY(c, hipStreamWaitEvent(c->side, c->side_fork, 0));
		const LaunchGeom g = make_geom(c->W, c->H, false);
		LaunchArgs a;
		a.tiles_x = g.tiles_x; a.tiles = g.tiles; a.rows = g.rows; a.half = 0; a.colour = 0; a.iter = 0;
		if (gen_candidates_all_views(c->d)) hipLaunchKernelGGL(dvp_gen_candidates_views, dim3(g.grid(), (unsigned)((c->NI - 1 + kCandGroup - 1) / kCandGroup)), dim3(256), 0, c->side, c->d, a);
		else hipLaunchKernelGGL(dvp_gen_candidates, dim3(g.grid(), (unsigned)(c->NI - 1)), dim3(256), 0, c->side, c->d, a);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipEventRecord(c->side_join, c->side));
	}
	if (launch_stage(c, DVP_ST_GEN_EDGE_INFORM, 0, 0, true)) return 1;
	if (launch_stage(c, DVP_ST_FIND_NEAREST_STRONG, 0, 0)) return 1;
	if (launch_stage(c, DVP_ST_GEN_NEIGHBOURS, 0, 0)) return 1;
	if (launch_stage(c, DVP_ST_NEIGHBOUR_UPDATE, 0, 0)) return 1;
	if (side_work && !masked) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_join, 0));
	if (masked) {
		// the anchors are known now; RandomInitialization is the first to rewrite the selected views the candidates read: they
		// get a snapshot and run beside RandomInit and the first strong updates, joined in front of the first weak update
		if (!c->sector_taps) { c->error = "dvp_set_params must be called before running kernels"; return 1; }
		HIP_TRY(c, hipMemcpyAsync(c->sel_snap, c->selected_views, (c->L + c->W) * 4, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(c, hipEventRecord(c->side_fork, c->stream));
		HIP_TRY(c, hipStreamWaitEvent(c->side, c->side_fork, 0));
		HIP_TRY(c, hipMemsetAsync(c->cand_mask, 0, c->L, c->side));
		HIP_TRY(c, hipMemsetAsync(c->cand_n, 0, 4, c->side));
		ListArgs la;
		la.base = 0; la.count = c->d.weak_black + c->d.weak_red; la.iter = 0; la.covered_rows = 0; la.group = 1; la.run = 1;
		if (la.count > 0) hipLaunchKernelGGL(dvp_anchor_mask, dim3((la.count + 255) / 256), dim3(256), 0, c->side, c->d, la, c->cand_mask);
		hipLaunchKernelGGL(dvp_mask_compact, dim3((unsigned)((c->L + 255) / 256)), dim3(256), 0, c->side, c->cand_mask, c->L, c->cand_list, c->cand_n);
		// (the list's length stays on the device: the grid covers the at most 11 anchors per WEAK pixel, blocks past the end leave)
		const size_t most = std::min(c->L, (size_t)la.count * (DVP_NEIGHBOUR_NUM - 1));
		Dev dsnap = c->d;
		dsnap.selected_views = c->sel_snap;
		if (most > 0 && gen_candidates_all_views(dsnap)) hipLaunchKernelGGL(dvp_gen_candidates_views_list, dim3((unsigned)((most + 255) / 256), (unsigned)((c->NI - 1 + kCandGroup - 1) / kCandGroup)), dim3(256), 0, c->side, dsnap, c->cand_list, c->cand_n);
		else if (most > 0) hipLaunchKernelGGL(dvp_gen_candidates_list, dim3((unsigned)((most + 255) / 256), (unsigned)(c->NI - 1)), dim3(256), 0, c->side, dsnap, c->cand_list, c->cand_n);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipEventRecord(c->side_join, c->side));
	}
	if (launch_stage(c, DVP_ST_RANDOM_INIT, 0, 0)) return 1;
	HIP_TRY(c, hipEventRecord(itl.a, c->stream));
	for (int i = 0; i < c->d.params.max_iterations; ++i) {
		if (launch_stage(c, DVP_ST_STRONG_UPDATE, i, 0)) return 1;
		if (launch_stage(c, DVP_ST_STRONG_UPDATE, i, 1)) return 1;
		// without WEAK pixels the three weak-path launches of an iteration do nothing but RANSACToGetFitPlane's copy
		// fit plane = plane (APD.cu:4208-4211), which only the last iteration's launch leaves behind
		if (c->d.weak_count == 0 && i == c->d.params.max_iterations - 1)
			HIP_TRY(c, hipMemcpyAsync(c->fit_planes, c->planes, c->L * 16, hipMemcpyDeviceToDevice, c->stream));
		if (c->d.weak_count > 0) {   // these three only touch WEAK pixels
			if (masked && i == 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_join, 0));   // the anchors' candidate records
			if (launch_stage(c, DVP_ST_RANSAC_FIT, i, 0)) return 1;
			// Black then red (APD.cu:4487-4489).  A WEAK pixel's update reads other pixels' state only at its anchors, which are STRONG
			// (GenNeighbours) and which no weak update writes: the two launches commute, and as the eight launches of the phased form
			// they are issued once over the whole WEAK list (half the launches and their tails).  dvp_run_stage keeps the colours apart.
			if (c->weak_phased && !c->anchor_tab_off && !getenv("DVP_WEAK_SPLIT_COLOURS")) {
				if (launch_stage(c, DVP_ST_WEAK_UPDATE, i, 2)) return 1;
			} else {
				if (launch_stage(c, DVP_ST_WEAK_UPDATE, i, 0)) return 1;
				if (launch_stage(c, DVP_ST_WEAK_UPDATE, i, 1)) return 1;
			}
		}
	}
	if (masked && c->d.params.max_iterations <= 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->side_join, 0));
	HIP_TRY(c, hipEventRecord(itl.b, c->stream));
	if (launch_stage(c, DVP_ST_GET_DEPTH_NORMAL, 0, 0)) return 1;
	if (launch_stage(c, DVP_ST_FILTER_STRONG, 0, 0)) return 1;
	if (launch_stage(c, DVP_ST_FILTER_STRONG, 0, 1)) return 1;
	// DepthToWeak and LocalRefine (APD.cu:4502-4505) in ONE launch: LocalRefine's sweep repeats eleven planes of
	// DepthToWeak's (dvp_strong.hpp, depth_to_weak_px).  Timed in the DVP_ST_DEPTH_TO_WEAK bucket.
	if (launch_stage(c, DVP_ST_DEPTH_TO_WEAK, 0, 0, true)) return 1;
	HIP_TRY(c, hipEventRecord(tot.b, c->stream));
	c->events.push_back(tot);
	c->events.push_back(itl);
	return 0;
}

// ---- results ----------------------------------------------------------------------------------
int dvp_download_state(dvp_ctx* c, float* planes, uint32_t* views, uint8_t* weak, int32_t* radius) {
	if (set_device(c)) return 1;
	const size_t L = c->L;
	if (planes) HIP_TRY(c, hipMemcpyAsync(planes, c->planes, L * 16, hipMemcpyDeviceToHost, c->stream));
	if (views) HIP_TRY(c, hipMemcpyAsync(views, c->selected_views, L * 4, hipMemcpyDeviceToHost, c->stream));
	if (weak) HIP_TRY(c, hipMemcpyAsync(weak, c->weak_info, L, hipMemcpyDeviceToHost, c->stream));
	if (radius) HIP_TRY(c, hipMemcpyAsync(radius, c->radius, L * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}

static void download_done(dvp_ctx* c) {
	{ std::lock_guard<std::mutex> lk(c->dl_m); c->dl_busy = false; c->dl_fetching = false; }
	c->dl_cv.notify_all();
}
int dvp_download_maps_begin(dvp_ctx* c, float* depth_device_copy) {
	if (set_device(c)) return 1;
	{   // the previous view's maps must have been fetched: its background job may not even have started yet, so this waits —
		// but not for ever (a second _begin after a _finish that never comes would otherwise hang the driver thread)
		int limit_s = 120;
		if (const char* e = getenv("DVP_DOWNLOAD_WAIT_S")) limit_s = atoi(e);
		std::unique_lock<std::mutex> lk(c->dl_m);
		if (!c->dl_cv.wait_for(lk, std::chrono::seconds(limit_s > 0 ? limit_s : 1), [c] { return !c->dl_busy; })) {
			c->error = "dvp_download_maps_begin: the maps staged by the previous dvp_download_maps_begin were never fetched (dvp_download_maps_finish)";
			return 1;
		}
		c->dl_busy = true;
	}
	const size_t L = c->L;
	auto fail = [c]() { download_done(c); return 1; };
	if (!c->maps_out && dalloc(c, &c->maps_out, L * 25, false)) return fail();
	float* d_depth = reinterpret_cast<float*>(c->maps_out);
	float* d_normal = d_depth + L;
	uint8_t* d_state = c->maps_out + L * 16;
	hipLaunchKernelGGL(dvp_unpack_maps, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream, c->planes, c->weak_info, L, c->d.params.depth_min, c->d.params.depth_max,
	                   d_depth, d_normal, d_state);
	if (hipGetLastError() != hipSuccess) { c->error = "dvp_download_maps_begin: launch failed"; return fail(); }
	// the view maps and the radius map are live state: the next view's uploads overwrite them
	if (hipMemcpyAsync(c->maps_out + L * 17, c->selected_views, L * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
	    hipMemcpyAsync(c->maps_out + L * 21, c->radius, L * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
	    (depth_device_copy && hipMemcpyAsync(depth_device_copy, d_depth, L * 4, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) ||
	    hipStreamSynchronize(c->stream) != hipSuccess) { c->error = "dvp_download_maps_begin: device copies failed"; return fail(); }
	return 0;
}
int dvp_download_maps_finish(dvp_ctx* c, float* depth, float* normal_xyz, uint32_t* views, uint8_t* weak, int32_t* radius) {
	t_finish_error.c = c;
	t_finish_error.msg.clear();
	{
		std::lock_guard<std::mutex> lk(c->dl_m);
		if (!c->dl_busy || c->dl_fetching) { t_finish_error.msg = "dvp_download_maps_finish without dvp_download_maps_begin"; return 1; }
		c->dl_fetching = true;
	}
	auto fail = [c](const char* what) { t_finish_error.msg = what; download_done(c); return 1; };
	if (hipSetDevice(c->device) != hipSuccess) return fail("hipSetDevice failed");
	if (!depth || !normal_xyz || !weak) return fail("dvp_download_maps: depth, normal and weak_info are required");
	if (!c->copy && hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
	const size_t L = c->L;
	// One DMA into pinned memory, then plain copies into the caller's (pageable) maps.  Copies straight into pageable memory go
	// through the runtime's staging path: at 6208x4128 they took 45-50 ms instead of 13 and the driver thread's own uploads for
	// the next view waited behind them (profiles/r05_ab_notes.txt).
	if (!c->maps_host && hipHostMalloc(reinterpret_cast<void**>(&c->maps_host), L * 25, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->maps_host = nullptr; }
	if (c->maps_host) {
		if (hipMemcpyAsync(c->maps_host, c->maps_out, L * 25, hipMemcpyDeviceToHost, c->copy) != hipSuccess || hipStreamSynchronize(c->copy) != hipSuccess)
			return fail("dvp_download_maps_finish: copy to the host failed");
		struct Part { void* dst; size_t off, bytes; };
		const Part parts[5] = { { depth, 0, L * 4 }, { normal_xyz, L * 4, L * 12 }, { weak, L * 16, L }, { views, L * 17, L * 4 }, { radius, L * 21, L * 4 } };
		std::vector<std::thread> team;
		const uint8_t* src = c->maps_host;
		for (const Part& pt : parts) {
			if (!pt.dst) continue;
			const size_t chunk = (size_t)32 << 20;
			if (pt.bytes <= chunk) { std::memcpy(pt.dst, src + pt.off, pt.bytes); continue; }
			const int nt = (int)std::min<size_t>(4, (pt.bytes + chunk - 1) / chunk);
			for (int t = 0; t < nt; ++t) {
				const size_t b0 = pt.bytes * t / nt, b1 = pt.bytes * (t + 1) / nt;
				team.emplace_back([=]() { std::memcpy(static_cast<uint8_t*>(pt.dst) + b0, src + pt.off + b0, b1 - b0); });
			}
		}
		for (std::thread& t : team) t.join();
		download_done(c);
		return 0;
	}
	const uint8_t* m = c->maps_out;
	if (hipMemcpyAsync(depth, m, L * 4, hipMemcpyDeviceToHost, c->copy) != hipSuccess ||
	    hipMemcpyAsync(normal_xyz, m + L * 4, L * 12, hipMemcpyDeviceToHost, c->copy) != hipSuccess ||
	    hipMemcpyAsync(weak, m + L * 16, L, hipMemcpyDeviceToHost, c->copy) != hipSuccess ||
	    (views && hipMemcpyAsync(views, m + L * 17, L * 4, hipMemcpyDeviceToHost, c->copy) != hipSuccess) ||
	    (radius && hipMemcpyAsync(radius, m + L * 21, L * 4, hipMemcpyDeviceToHost, c->copy) != hipSuccess) ||
	    hipStreamSynchronize(c->copy) != hipSuccess) return fail("dvp_download_maps_finish: copies to the host failed");
	download_done(c);
	return 0;
}
int dvp_download_maps(dvp_ctx* c, float* depth, float* normal_xyz, uint32_t* views, uint8_t* weak, int32_t* radius) {
	if (!depth || !normal_xyz || !weak) { c->error = "dvp_download_maps: depth, normal and weak_info are required"; return 1; }
	if (dvp_download_maps_begin(c, nullptr)) return 1;
	return dvp_download_maps_finish(c, depth, normal_xyz, views, weak, radius);
}

static void* buffer_ptr(dvp_ctx* c, int id, size_t* bytes) {
	const size_t L = c->L, S = (size_t)c->NI - 1;
	const size_t wc = c->d.weak_count > 0 ? (size_t)c->d.weak_count : 1;
	switch (id) {
	case DVP_BUF_PLANES: *bytes = L * 16; return c->planes;
	case DVP_BUF_COSTS: *bytes = L * 4; return c->costs;
	case DVP_BUF_SELECTED_VIEWS: *bytes = L * 4; return c->selected_views;
	case DVP_BUF_VIEW_WEIGHT: *bytes = L * 32; return c->view_weight;
	case DVP_BUF_WEAK_INFO: *bytes = L; return c->weak_info;
	case DVP_BUF_WEAK_RELIABLE: *bytes = L; return c->weak_reliable;
	case DVP_BUF_WEAK_NEAREST_STRONG: *bytes = L * 4; return c->weak_nearest_strong;
	case DVP_BUF_NEIGHBOURS_MAP: *bytes = L * 4; return c->neighbours_map;
	case DVP_BUF_NEIGHBOURS: *bytes = wc * DVP_NEIGHBOUR_NUM * 4; return c->neighbours;
	case DVP_BUF_FIT_PLANES: *bytes = L * 16; return c->fit_planes;
	case DVP_BUF_CANDIDATE: *bytes = L * S * 8 * 4; return c->candidate;
	case DVP_BUF_EDGE: *bytes = L; return c->edge;
	case DVP_BUF_EDGE_NEIGH: *bytes = L * 8 * 4; return c->edge_neigh;
	case DVP_BUF_LABEL: *bytes = L * 4; return c->label;
	case DVP_BUF_LABEL_BOUNDARY: *bytes = wc * 8 * 4; return c->label_boundary;
	case DVP_BUF_COMPLEX: *bytes = wc * 4; return c->complex_;
	case DVP_BUF_RADIUS: *bytes = L * 4; return c->radius;
	}
	*bytes = 0;
	return nullptr;
}
long long dvp_buffer_bytes(dvp_ctx* c, int id) { size_t b; buffer_ptr(c, id, &b); return (long long)b; }
// DVP_BUF_CANDIDATE is [view][pixel][8] on the device and [pixel][view][8] (the reference's order,
// main.h:41 with the view stride fixed) at the boundary
static void candidate_transpose(const dvp_ctx* c, const s2* src, s2* dst, bool to_device) {
	const size_t L = c->L, S = (size_t)c->NI - 1;
	for (size_t v = 0; v < S; ++v)
		for (size_t p = 0; p < L; ++p) {
			const size_t dev = (v * L + p) * 8, host = (p * S + v) * 8;
			std::memcpy(to_device ? dst + dev : dst + host, to_device ? src + host : src + dev, 8 * sizeof(s2));
		}
}
int dvp_download_buffer(dvp_ctx* c, int id, void* dst) {
	if (set_device(c)) return 1;
	size_t b; void* p = buffer_ptr(c, id, &b);
	if (!p) { c->error = "bad or unallocated buffer id"; return 1; }
	if (id == DVP_BUF_CANDIDATE) {
		std::vector<s2> tmp(b / sizeof(s2));
		HIP_TRY(c, hipMemcpyAsync(tmp.data(), p, b, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		candidate_transpose(c, tmp.data(), (s2*)dst, false);
		return 0;
	}
	HIP_TRY(c, hipMemcpyAsync(dst, p, b, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}
int dvp_upload_buffer(dvp_ctx* c, int id, const void* src) {
	c->anchor_tab_valid = false;
	if (set_device(c)) return 1;
	size_t b; void* p = buffer_ptr(c, id, &b);
	if (!p) { c->error = "bad or unallocated buffer id"; return 1; }
	std::vector<s2> tmp;
	if (id == DVP_BUF_CANDIDATE) {
		tmp.resize(b / sizeof(s2));
		candidate_transpose(c, (const s2*)src, tmp.data(), true);
		src = tmp.data();
	}
	HIP_TRY(c, hipMemcpyAsync(p, src, b, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return 0;
}
int dvp_weak_count(dvp_ctx* c) { return c->d.weak_count; }

int dvp_get_timings(dvp_ctx* c, DvpTimings* out) {
	if (set_device(c)) return 1;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	for (auto& e : c->events) {
		float ms = 0.0f;
		HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
		if (e.stage == EV_TOTAL) c->timings.total_ms += ms;
		else if (e.stage == EV_ITER_LOOP) c->timings.iter_loop_ms += ms;
		else { c->timings.stage_ms[e.stage] += ms; c->timings.stage_launches[e.stage] += 1; }
		(void)hipEventDestroy(e.a);
		(void)hipEventDestroy(e.b);
	}
	c->events.clear();
	if (out) *out = c->timings;
	return 0;
}
int dvp_reset_timings(dvp_ctx* c) {
	if (dvp_get_timings(c, nullptr)) return 1;
	std::memset(&c->timings, 0, sizeof(DvpTimings));
	return 0;
}

// ---- KAT / micro-benchmark ---------------------------------------------------------------------
namespace {
struct DevBuf {   // hipMalloc'ed scratch released on every exit path
	void* p = nullptr;
	~DevBuf() { if (p) (void)hipFree(p); }
};
struct EventPairGuard {
	hipEvent_t a = nullptr, b = nullptr;
	~EventPairGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
}  // namespace

int dvp_eval_cost_vectors(dvp_ctx* c, const int32_t* px, const float* planes, int n, float* out, float* kernel_ms) {
	if (set_device(c)) return 1;
	if (!c->sector_taps) { c->error = "dvp_set_params must be called first"; return 1; }
	if (n <= 0) return 0;
	const size_t S = (size_t)c->NI - 1;
	DevBuf dpx, dpl, dout;
	EventPairGuard ev;
	HIP_TRY(c, hipMalloc(&dpx.p, (size_t)n * 8));
	HIP_TRY(c, hipMalloc(&dpl.p, (size_t)n * 16));
	HIP_TRY(c, hipMalloc(&dout.p, (size_t)n * S * 4));
	HIP_TRY(c, hipMemcpyAsync(dpx.p, px, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(dpl.p, planes, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipEventCreate(&ev.a));
	HIP_TRY(c, hipEventCreate(&ev.b));
	HIP_TRY(c, hipEventRecord(ev.a, c->stream));
	hipLaunchKernelGGL(dvp_cost_vectors, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d, (const int*)dpx.p, (const f4*)dpl.p, n, (float*)dout.p);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventRecord(ev.b, c->stream));
	HIP_TRY(c, hipMemcpyAsync(out, dout.p, (size_t)n * S * 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (kernel_ms) HIP_TRY(c, hipEventElapsedTime(kernel_ms, ev.a, ev.b));
	return 0;
}

int dvp_bench_cost_kernel(dvp_ctx* c, int repeat, float* mean_kernel_ms, uint64_t* evals_per_launch) {
	if (set_device(c)) return 1;
	if (!c->sector_taps) { c->error = "dvp_set_params must be called first"; return 1; }
	if (repeat < 1) repeat = 1;
	const LaunchGeom g = make_geom(c->W, c->H, false);
	LaunchArgs a;
	a.tiles_x = g.tiles_x; a.tiles = g.tiles; a.rows = g.rows; a.half = 0; a.colour = 0; a.iter = 0;
	EventPairGuard ev;
	HIP_TRY(c, hipEventCreate(&ev.a));
	HIP_TRY(c, hipEventCreate(&ev.b));
	hipLaunchKernelGGL(dvp_cost_all_pixels, dim3(g.grid()), dim3(256), 0, c->stream, c->d, a, c->scratch_out);   // warm-up
	HIP_TRY(c, hipEventRecord(ev.a, c->stream));
	for (int i = 0; i < repeat; ++i)
		hipLaunchKernelGGL(dvp_cost_all_pixels, dim3(g.grid()), dim3(256), 0, c->stream, c->d, a, c->scratch_out);
	HIP_TRY(c, hipEventRecord(ev.b, c->stream));
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	float ms = 0.0f;
	HIP_TRY(c, hipEventElapsedTime(&ms, ev.a, ev.b));
	if (mean_kernel_ms) *mean_kernel_ms = ms / repeat;
	if (evals_per_launch) *evals_per_launch = (uint64_t)c->L * (uint64_t)(c->NI - 1);
	return 0;
}

#ifdef DVP_PROBE
int dvp_probe(dvp_ctx* c, int mode, int K, int stride, int repeat, float* mean_ms, float* checksum) {
	if (set_device(c)) return 1;
	const LaunchGeom g = make_geom(c->W, c->H, false);
	LaunchArgs a;
	a.tiles_x = g.tiles_x; a.tiles = g.tiles; a.rows = g.rows; a.half = 0; a.colour = 0; a.iter = 0;
	const int S = c->NI - 1, P = 64 / S;
	const int waves = ((c->W + P - 1) / P) * c->H;
	EventPairGuard ev;
	HIP_TRY(c, hipEventCreate(&ev.a));
	HIP_TRY(c, hipEventCreate(&ev.b));
	for (int i = 0; i <= repeat; ++i) {
		if (i == 1) HIP_TRY(c, hipEventRecord(ev.a, c->stream));
		HIP_TRY(c, hipMemsetAsync(c->scratch_out, 0, c->L * 4, c->stream));
		if (mode == 0) hipLaunchKernelGGL(dvp_probe_a, dim3(g.grid()), dim3(256), 0, c->stream, c->d, a, c->scratch_out, K, stride);
		else if (mode == 1) hipLaunchKernelGGL(dvp_probe_b, dim3((waves + 3) / 4), dim3(256), 0, c->stream, c->d, c->scratch_out, K, stride);
		else if (mode == 2) hipLaunchKernelGGL(dvp_probe_c, dim3(g.grid()), dim3(256), 0, c->stream, c->d, a, c->scratch_out, (unsigned)K);   // K = view mask
		else hipLaunchKernelGGL(dvp_probe_d, dim3((unsigned)(g.tiles_x * c->H * (64 / DVP_PROBE_NPX))), dim3(64), 0, c->stream, c->d, c->scratch_out, (unsigned)K);
	}
	HIP_TRY(c, hipEventRecord(ev.b, c->stream));
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	float ms = 0.0f;
	HIP_TRY(c, hipEventElapsedTime(&ms, ev.a, ev.b));
	*mean_ms = ms / repeat;
	std::vector<float> h(c->L);
	HIP_TRY(c, hipMemcpy(h.data(), c->scratch_out, c->L * 4, hipMemcpyDeviceToHost));
	double sum = 0;
	for (float v : h) sum += v;
	*checksum = (float)(sum / (double)c->L);
	return 0;
}
#endif

}  // extern "C"
