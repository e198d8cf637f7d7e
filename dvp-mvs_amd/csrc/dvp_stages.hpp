// dvp_stages.hpp — launch geometry shared by every kernel, per-stage pixel dispatch, and the
// host-side uniform constants (sector table, anchor-search rotation constants).
#ifndef DVP_STAGES_HPP_
#define DVP_STAGES_HPP_

#include "dvp_weak_wave.hpp"
#include "dvp_weak_phased.hpp"
#include <vector>
#include <cmath>

namespace dvp {

// Tile = 64 x 4 lanes: a wave is 64 x-adjacent pixels (coalesced state reads, near-contiguous
// source gathers), a 256-thread workgroup is 4 such rows.
//  * full launches: pixel (tile_x*64 + lane, tile_y*4 + wave)
//  * half (red/black) launches: wave = one row PAIR; lane l sits at x = tile_x*64 + l and
//    y = 2*pair + ((x&1) ^ colour) — the reference's pixel map (APD.cu:3093-3100).  The reference
//    grid covers 16*ceil((H/2)/16) row pairs (APD.cu:4421-4424): for odd H with (H/2)%16 == 0 the
//    last row is never updated; `rows()` reproduces that.
struct LaunchGeom {
	int tiles_x, tiles_y, tiles;
	int rows;                             // rows (full) or row pairs (half) covered
	bool half;
	int grid() const { return tiles; }
};
inline LaunchGeom make_geom(int W, int H, bool half) {
	LaunchGeom g;
	g.half = half;
	g.rows = half ? ((H / 2) + 15) / 16 * 16 : H;
	g.tiles_x = (W + 63) / 64;
	g.tiles_y = (g.rows + 3) / 4;
	g.tiles = g.tiles_x * g.tiles_y;
	return g;
}

// XCD-aware block -> tile map.  Workgroup b is dispatched to XCD b % 8 (observed placement, a
// speed matter only).  Tiles are traversed in vertical strips of 8 tile columns: inside a strip
// XCD j owns column j and walks it top to bottom, so its 64 resident tiles form a 64 x 512-pixel
// block whose source rows fit its 4 MiB L2, and the eight XCDs work side by side on the same image
// rows (shared in the Infinity Cache).  The ragged last strip is dealt row-major over all XCDs.
// Measured (3104x2064, S=5, ms per strong-update launch): one contiguous band of the image per XCD
// 22.8, plain row-major 20.4, this map 19.3.
DVP_HD bool block_to_pixel(int block, int lane, int wave, int tiles_x, int tiles, int rows, int half, int colour,
	int W, int H, int* px, int* py) {
	if (block >= tiles) return false;
	const int tiles_y = tiles / tiles_x;
	const int per_strip = 8 * tiles_y;
	const int st = block / per_strip;
	const int rem = block - st * per_strip;
	const int w_last = tiles_x - st * 8;             // columns left in this strip
	int ty, tx;
	if (w_last >= 8) {
		ty = rem >> 3;
		tx = st * 8 + (rem & 7);
	} else {
		ty = rem / w_last;
		tx = st * 8 + (rem - ty * w_last);
	}
	const int x = tx * 64 + lane;
	const int r = ty * 4 + wave;
	if (x >= W || r >= rows) return false;
	int y = r;
	if (half) y = 2 * r + ((x & 1) ^ colour);
	if (y >= H) return false;
	*px = x;
	*py = y;
	return true;
}

// per-pixel predicate + body of launch site `STAGE` (DVP_ST_*), APD.cu:3091-3165, 3296-3328
constexpr int kNarrowViews = 8;   // view capacity of the narrow strong-update instantiation
// internal launch site: DepthToWeak with LocalRefine done by the same thread (dvp_run_patchmatch issues the two
// back to back; depth_to_weak_px<SMP, true>).  dvp_run_stage keeps the two separate launches.
constexpr int kStageSweeps = 100;
constexpr int kSweepBorderOnly = -1;
// internal launch sites of the split strong update (dvp_strong.hpp: strong_eval_px / strong_decide_px / strong_refine_px)
constexpr int kStageStrongEval = 101, kStageStrongRefine = 102, kStageStrongRefineLanes = 103;   // 103: every lane walks its own (hypothesis, view) sequence
template <int STAGE, int SMP, int MV = 32>
DVP_HD void run_pixel(const Dev& d, int px, int py, int iter, unsigned long long* nevals, PatchTab tab) {
	const int center = px + py * d.width;
	if (STAGE == DVP_ST_GEN_EDGE_INFORM) {
		gen_edge_inform_px(d, px, py);
#if !defined(__HIP_DEVICE_COMPILE__)   // device: own launch shape, pixels x views (dvp_gen_candidates)
		if (gen_candidates_all_views(d)) { for (int v0 = 0; v0 < d.params.num_images - 1; v0 += kCandGroup) gen_candidates_views_px<kCandGroup>(d, px, py, v0); }   // (as the engine chooses)
		else for (int v = 0; v < d.params.num_images - 1; ++v) gen_candidates_px(d, px, py, v);
#endif
	}
	else if (STAGE == DVP_ST_FIND_NEAREST_STRONG) find_nearest_strong_px(d, px, py);
	else if (STAGE == DVP_ST_GEN_NEIGHBOURS) {
		// device: the directional search one lane per WEAK pixel (dvp_gen_neighbours_list) or, DVP_GN_WAVE=1, one wave per
		// WEAK pixel (dvp_gen_neighbours_search); label extension + fit one wave per WEAK pixel (dvp_gen_neighbours_fit).
		// Host emulation: the same choice.
#if !defined(__HIPCC__)
		if (d.weak_info[center] == DVP_WEAK) {
			const char* gw_env = getenv("DVP_GN_WAVE");
			if (gw_env && atoi(gw_env) != 0) { GnShared gs; gen_neighbours_search_wave(d, px, py, gs); }
			else { s2 pts[kGnDirSlots]; gen_neighbours_px(d, px, py, pts, 1); }
			FitShared sh;
			gen_neighbours_fit_wave(d, px, py, sh);
		}
#endif
	}
	else if (STAGE == DVP_ST_NEIGHBOUR_UPDATE) neighbour_update_px(d, px, py);
	else if (STAGE == DVP_ST_RANDOM_INIT) random_init_px<SMP>(d, px, py, tab, nevals);
	else if (STAGE == DVP_ST_STRONG_UPDATE) { if (d.weak_info[center] != DVP_WEAK) strong_update_px<SMP, MV>(d, px, py, tab, iter, nevals); }
	else if (STAGE == kStageStrongEval) { if (d.weak_info[center] != DVP_WEAK) strong_eval_px<SMP>(d, px, py, tab, nevals); }
	else if (STAGE == kStageStrongRefine) { if (d.weak_info[center] != DVP_WEAK) strong_refine_px<SMP>(d, px, py, tab, nevals); }
	else if (STAGE == kStageStrongRefineLanes) { if (d.weak_info[center] != DVP_WEAK) strong_refine_px<SMP, true>(d, px, py, tab, nevals); }
	else if (STAGE == DVP_ST_RANSAC_FIT) {
		// device: one lane per WEAK pixel, or (DVP_RANSAC_WAVE=1) one wave per WEAK pixel (dvp_ransac_fit_plane_wave); the host emulation follows the switch
#if !defined(__HIPCC__)
		const char* rw = getenv("DVP_RANSAC_WAVE");
		if (rw && atoi(rw) != 0) { RansacShared sh; ransac_fit_plane_wave(d, px, py, iter, sh); }
		else
#endif
		ransac_fit_plane_px(d, px, py, iter);
	}
	else if (STAGE == DVP_ST_WEAK_UPDATE) {
		// device: own launch shape (one wave per WEAK pixel, dvp_weak_update_wave); this branch is the
		// host emulation of that wave (tests/emul): DVP_LANES loops over the 64 lanes
#if !defined(__HIPCC__)
		if (d.weak_info[center] == DVP_WEAK) {
			if (d.anchor_tab) {
				WeakSharedT<1> sh;
				if (d.images8) weak_update_wave<SMP, 1, 1>(d, px, py, iter, nevals, sh);
				else weak_update_wave<SMP, 0, 1>(d, px, py, iter, nevals, sh);
			} else {
				WeakSharedT<0> sh;
				if (d.images8) weak_update_wave<SMP, 1, 0>(d, px, py, iter, nevals, sh);
				else weak_update_wave<SMP, 0, 0>(d, px, py, iter, nevals, sh);
			}
		}
#endif
	}
	else if (STAGE == DVP_ST_GET_DEPTH_NORMAL) get_depth_normal_px(d, px, py);
	else if (STAGE == DVP_ST_FILTER_STRONG) { if (d.weak_info[center] != DVP_WEAK) filter_strong_px(d, px, py); }
	else if (STAGE == DVP_ST_DEPTH_TO_WEAK) depth_to_weak_px<SMP, false>(d, px, py, tab, nevals);
	else if (STAGE == kStageSweeps) {
		// iter == kSweepBorderOnly: the border launch of the view-compacted form (sweep_* in dvp_strong.hpp), which leaves the
		// 6-pixel frame — UNKNOWN for DepthToWeak, a full LocalRefine — to this kernel
		if (iter == kSweepBorderOnly && !sweep_is_border(d, px, py)) return;
		depth_to_weak_px<SMP, true>(d, px, py, tab, nevals);
	}
	else if (STAGE == DVP_ST_LOCAL_REFINE) local_refine_px<SMP>(d, px, py, tab, nevals);
}
// launch sites whose kernels need the per-lane patch table (LDS on the GPU)
constexpr bool stage_uses_tab(int stage) {
	return stage == DVP_ST_RANDOM_INIT || stage == DVP_ST_STRONG_UPDATE || stage == DVP_ST_DEPTH_TO_WEAK || stage == DVP_ST_LOCAL_REFINE || stage == kStageSweeps ||
	       stage == kStageStrongEval || stage == kStageStrongRefine;
}
constexpr bool stage_is_half_c(int stage) {
	return stage == DVP_ST_STRONG_UPDATE || stage == DVP_ST_WEAK_UPDATE || stage == DVP_ST_FILTER_STRONG || stage == kStageStrongEval || stage == kStageStrongRefine;
}
inline bool stage_is_half(int stage) {
	return stage == DVP_ST_STRONG_UPDATE || stage == DVP_ST_WEAK_UPDATE || stage == DVP_ST_FILTER_STRONG;
}

// ---- host-side uniform constants ---------------------------------------------------------------
// The window offsets of GenEdgeInform grouped by 30-degree sector, exactly as calculateAngle/getRegion
// bin them (APD.cu:797-821; the angle passes through a float, APD.cu:3759), each sector in the
// reference's visit order (i outer, j inner).  taps[k] = (i + radius) | (j + radius) << 16,
// start[s]..start[s+1] = sector s.  Offsets that fall in no sector are dropped like getRegion's -1.
inline void make_sector_taps(int radius, std::vector<int>* taps, std::vector<int>* start) {
	const double kPI = 3.14159265358979323846;   // APD.h:6
	std::vector<std::vector<int>> by(12);
	for (int i = -radius; i <= radius; ++i)
		for (int j = -radius; j <= radius; ++j) {
			if (i == 0 && j == 0) continue;
			double deg = std::atan2((double)j, (double)i) * (180.0 / kPI);
			if (deg < 0) deg += 360.0;
			const double a = (double)(float)deg;
			for (int q = 0; q < 12; ++q)
				if (a >= 30.0 * q && a < 30.0 * (q + 1)) by[q].push_back((i + radius) | ((j + radius) << 16));
		}
	taps->clear();
	start->assign(13, 0);
	for (int q = 0; q < 12; ++q) {
		(*start)[q] = (int)taps->size();
		taps->insert(taps->end(), by[q].begin(), by[q].end());
	}
	(*start)[12] = (int)taps->size();
}
// GenNeighbours' per-launch constants (APD.cu:3375-3380), evaluated in double like the reference
inline void set_neighbour_consts(Dev* d) {
	const double kPI = 3.14159265358979323846;
	const int rt = d->params.rotate_time > 0 ? d->params.rotate_time : 1;
	const float angle = 45.0f / rt;
	d->nb_cos = (float)std::cos(angle * kPI / 180.f);
	d->nb_sin = (float)std::sin(angle * kPI / 180.f);
	d->nb_thresh = (float)std::cos((angle / 2.0f) * kPI / 180.0f);
	const int sr = (int)(std::tan((angle / 2.0f) * kPI / 180.0f) * 20);
	d->nb_shift_range = sr < 1 ? 1 : sr;
}

}  // namespace dvp
#endif
