// dvp_strong.hpp — per-pixel bodies of the strong path and the post-processing kernels.
// One lane owns one pixel; a wave owns 64 x-adjacent pixels so that, once planes have locally
// converged, the 64 bilinear gathers of a tap hit a handful of 128-B lines.
#ifndef DVP_STRONG_HPP_
#define DVP_STRONG_HPP_

#include "dvp_ncc.hpp"

namespace dvp {

DVP_HD void sort_small(float* v, int n) {   // insertion sort, APD.cu:114-123
	for (int i = 1; i < n; i++) {
		const float tmp = v[i];
		int j = i;
		for (; j >= 1 && tmp < v[j - 1]; j--) v[j] = v[j - 1];
		v[j] = tmp;
	}
}

// GenerateRandomNormal_YZL (APD.cu:501-588): rejection-sample a unit normal that faces the
// reference viewing ray and the (quirkily transformed) viewing rays of every selected source view.
// `sel` = the pixel's selected-view mask (selected_views[center] at the time of the call)
DVP_HD f4 random_normal_yzl_sel(const Dev& d, int px, int py, Rng& rng, float depth, uint32_t sel) {
	const int W = d.width, H = d.height;
	const DvpCamera rc = load_camera(d, 0);
	f3 vd[20];
	{
		const f4 v0 = view_direction(rc, px, py, depth);
		vd[0] = mk3(v0.x, v0.y, v0.z);
	}
	int index = 1;
	for (int v = 1; v < d.params.num_images; ++v) {
		if (!is_set(sel, v - 1)) continue;
		const DvpCamera sc = load_camera(d, v);
		const f3 fwd = point_on_world((float)px, (float)py, depth, rc);
		f2 sp;
		float sd;
		project_on_camera(fwd, sc, &sp, &sd);
		const float sx = fminf(fmaxf(sp.x, -32768.0f), 32767.0f);
		const float sy = fminf(fmaxf(sp.y, -32768.0f), 32767.0f);
		const int ix = (int)((float)(int)sx + 0.5f), iy = (int)((float)(int)sy + 0.5f);   // APD.cu:525
		float src_depth = 1.0f;   // reference leaves it uninitialised outside the image (APD.cu:526)
		if (d.params.geom_consistency) {
			if (ix >= 0 && ix < W && iy >= 0 && iy < H)
				src_depth = tex_texel(d.depths + (size_t)v * d.plane_stride, d.org, d.pitch, W, H, (int)sx, (int)sy);
		}
		const f4 dir = view_direction(sc, ix, iy, src_depth);
		// R_c = R_ref * R_src^T ; R_f = R_c * {x, y, x} with row 2 using R_c[7] twice (APD.cu:14-18, 540-544)
		float Rc[9];
		for (int i = 0; i < 3; ++i)
			for (int j = 0; j < 3; ++j) {
				float acc = 0.0f;
				for (int k = 0; k < 3; ++k) acc += rc.R[i * 3 + k] * sc.R[j * 3 + k];
				Rc[i * 3 + j] = acc;
			}
		const float b0 = dir.x, b1 = dir.y, b2 = dir.x;
		const float f0 = Rc[0] * b0 + Rc[1] * b1 + Rc[2] * b2;
		const float f1 = Rc[3] * b0 + Rc[4] * b1 + Rc[5] * b2;
		const float f2_ = Rc[6] * b0 + Rc[7] * b1 + Rc[7] * b2;
		const float norm = sqrtf(f0 * f0 + f1 * f1 + f2_ * f2_);
		if (index < 20) vd[index++] = mk3(f0 / norm, f1 / norm, f2_ / norm);
	}
	int times = 200;
	f4 n = mk4(0, 0, 0, 0);
	while (times > 0) {
		float q1 = 1.0f, q2 = 1.0f, s = 2.0f;
		while (s >= 1.0f) {
			q1 = 2.0f * rng.uniform() - 1.0f;
			q2 = 2.0f * rng.uniform() - 1.0f;
			s = q1 * q1 + q2 * q2;
		}
		const float sq = sqrtf(1.0f - s);
		n.x = 2.0f * q1 * sq;
		n.y = 2.0f * q2 * sq;
		n.z = 1.0f - 2.0f * s;
		bool ok = true;
		for (int i = 0; i < index; i++) {
			const float dp = n.x * vd[i].x + n.y * vd[i].y + n.z * vd[i].z;
			if (dp > 0.0f) { ok = false; break; }
		}
		if (ok) break;
		times--;
	}
	normalize3(&n);
	return n;
}

DVP_HD f4 random_normal_yzl(const Dev& d, int px, int py, Rng& rng, float depth) {
	return random_normal_yzl_sel(d, px, py, rng, depth, d.selected_views[py * d.width + px]);
}

// The same draw for the split strong update's decision launch (S <= MV <= 16 views, unrolled).  random_normal_yzl_sel walks the
// views with `if (!selected) continue`, fetches the source depth of each inside that loop and appends to vd[index++]: one dependent
// round trip per selected view and a dynamically indexed private array (scratch) that the rejection loop re-reads on every try —
// half of dvp_strong_decide's 8.6 ms at cfg3 (timing ablation, profiles/r06_ab_notes.txt 20).  Here: the depth texels of all views
// fetched together (clamped addresses, the value used only where the reference reads it), the directions in registers behind a
// mask.  Same operations per selected view, same random numbers; the order in which a candidate is tested against the directions
// does not matter (it has to face away from all of them), and with at most 16 views the reference's limit of 19 never binds.
template <int MV>
DVP_HD f4 random_normal_yzl_views(const Dev& d, int px, int py, Rng& rng, float depth, uint32_t sel) {
	static_assert(MV <= 16, "the reference keeps at most 19 source directions (APD.cu:546)");
	const int W = d.width, H = d.height;
	const int S = d.params.num_images - 1;
	const DvpCamera rc = load_camera(d, 0);
	f3 vd0;
	{
		const f4 v0 = view_direction(rc, px, py, depth);
		vd0 = mk3(v0.x, v0.y, v0.z);
	}
	const f3 fwd = point_on_world((float)px, (float)py, depth, rc);
	int ixs[MV], iys[MV], txs[MV], tys[MV];
	float dep[MV];
	uint32_t inside = 0;
#pragma unroll
	for (int v = 1; v <= MV; ++v) {
		const int vv = v <= S ? v : S;   // (a view that exists; beyond S the result is not used)
		const DvpCamera sc = load_camera(d, vv);
		f2 sp;
		float sd;
		project_on_camera(fwd, sc, &sp, &sd);
		const float sx = fminf(fmaxf(sp.x, -32768.0f), 32767.0f);
		const float sy = fminf(fmaxf(sp.y, -32768.0f), 32767.0f);
		const int ix = (int)((float)(int)sx + 0.5f), iy = (int)((float)(int)sy + 0.5f);   // APD.cu:525
		ixs[v - 1] = ix; iys[v - 1] = iy;
		txs[v - 1] = (int)sx; tys[v - 1] = (int)sy;
		if (ix >= 0 && ix < W && iy >= 0 && iy < H) inside |= 1u << (v - 1);
	}
	if (d.params.geom_consistency) {   // (without it there are no depth planes)
#pragma unroll
		for (int v = 1; v <= MV; ++v) {
			const int vv = v <= S ? v : S;
			dep[v - 1] = tex_texel(d.depths + (size_t)vv * d.plane_stride, d.org, d.pitch, W, H, txs[v - 1], tys[v - 1]);   // (clamped inside tex_texel)
		}
	} else {
#pragma unroll
		for (int v = 1; v <= MV; ++v) dep[v - 1] = 1.0f;
	}
	sched_fence();
	f3 vd[MV];
	uint32_t valid = 0;
#pragma unroll
	for (int v = 1; v <= MV; ++v) {
		const int vv = v <= S ? v : S;
		const DvpCamera sc = load_camera(d, vv);
		float src_depth = 1.0f;   // reference leaves it uninitialised outside the image (APD.cu:526)
		if (d.params.geom_consistency && ((inside >> (v - 1)) & 1u)) src_depth = dep[v - 1];
		const f4 dir = view_direction(sc, ixs[v - 1], iys[v - 1], src_depth);
		float Rc[9];
		for (int i = 0; i < 3; ++i)
			for (int j = 0; j < 3; ++j) {
				float acc = 0.0f;
				for (int k = 0; k < 3; ++k) acc += rc.R[i * 3 + k] * sc.R[j * 3 + k];
				Rc[i * 3 + j] = acc;
			}
		const float b0 = dir.x, b1 = dir.y, b2 = dir.x;
		const float f0 = Rc[0] * b0 + Rc[1] * b1 + Rc[2] * b2;
		const float f1 = Rc[3] * b0 + Rc[4] * b1 + Rc[5] * b2;
		const float f2_ = Rc[6] * b0 + Rc[7] * b1 + Rc[7] * b2;
		const float norm = sqrtf(f0 * f0 + f1 * f1 + f2_ * f2_);
		vd[v - 1] = mk3(f0 / norm, f1 / norm, f2_ / norm);
		if (v <= S && is_set(sel, v - 1)) valid |= 1u << (v - 1);
	}
	int times = 200;
	f4 n = mk4(0, 0, 0, 0);
	while (times > 0) {
		float q1 = 1.0f, q2 = 1.0f, s = 2.0f;
		while (s >= 1.0f) {
			q1 = 2.0f * rng.uniform() - 1.0f;
			q2 = 2.0f * rng.uniform() - 1.0f;
			s = q1 * q1 + q2 * q2;
		}
		const float sq = sqrtf(1.0f - s);
		n.x = 2.0f * q1 * sq;
		n.y = 2.0f * q2 * sq;
		n.z = 1.0f - 2.0f * s;
		bool ok = !(n.x * vd0.x + n.y * vd0.y + n.z * vd0.z > 0.0f);
#pragma unroll
		for (int i = 0; i < MV; i++) {
			const float dp = n.x * vd[i].x + n.y * vd[i].y + n.z * vd[i].z;
			if (((valid >> i) & 1u) && dp > 0.0f) ok = false;
		}
		if (ok) break;
		times--;
	}
	normalize3(&n);
	return n;
}

// RandomInitialization (APD.cu:1273-1309)
template <int SMP>
DVP_HD void random_init_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int center = py * d.width + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	f4 plane = d.planes[center];
	PatchCtx c;
	int radius, inc;
	patch_geometry(d, center, &radius, &inc);
	build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);

	if (P.state == DVP_FIRST_INIT) {
		if (plane.w > P.depth_max || plane.w < P.depth_min) {
			// GenerateRandomPlaneHypothesis_YZL (APD.cu:663-669)
			Rng rd(d.seed, (uint32_t)center, rng_site(PH_RANDOM_INIT, 0, SUB_DEPTH_RAND));
			Rng rn(d.seed, (uint32_t)center, rng_site(PH_RANDOM_INIT, 0, SUB_NORMAL));
			const float depth = rd.uniform() * (P.depth_max - P.depth_min) + P.depth_min;
			plane = random_normal_yzl(d, px, py, rn, depth);
			plane.w = distance_to_origin(rc, px, py, depth, plane);
		}
		d.planes[center] = plane;
		// ComputeMultiViewInitialCostandSelectedViews (APD.cu:1115-1161)
		float cv[32], cvs[32];
		int valid = 0;
		for (int v = 0; v < S; ++v) {
			const float cst = ncc_old<SMP>(d, c, px, py, v + 1, plane);
			cv[v] = cst;
			cvs[v] = cst;
			if (cst < 2.0f) valid++;
		}
		if (nevals) *nevals += (unsigned long long)S;
		sort_small(cvs, S);
		uint32_t sel = 0;
		const int top_k = DVP_MIN(valid, P.top_k);
		float cost = 2.0f;
		if (top_k > 0) {
			float acc = 0.0f;
			for (int i = 0; i < top_k; ++i) acc += cvs[i];
			const float thr = cvs[top_k - 1];
			for (int i = 0; i < S; ++i)
				if (cv[i] <= thr) set_bit(&sel, i);
			cost = acc / top_k;
		}
		d.selected_views[center] = sel;
		d.costs[center] = cost;
	} else {
		plane = normal_world_to_cam(rc, plane);
		const float depth = plane.w;
		plane.w = distance_to_origin(rc, px, py, depth, plane);
		d.planes[center] = plane;
		// ComputeMultiViewInitialCost (APD.cu:1163-1194)
		uint32_t sel = d.selected_views[center];
		int cnt = 0;
		float acc = 0.0f;
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			const float cst = ncc_old<SMP>(d, c, px, py, v + 1, plane);
			if (nevals) *nevals += 1;
			if (cst < 2.0f) { cnt++; acc += cst; }
			else unset_bit_ref(&sel, v);
		}
		d.selected_views[center] = sel;
		d.costs[center] = (cnt == 0) ? 2.0f : acc / cnt;
	}
}

// Multi-hypothesis joint view selection (APD.cu:2483-2530 == 2803-2850)
template <int MV = 32>   // MV: row stride of cost_array and capacity of priors / vw (>= number of source views)
DVP_HD void joint_view_selection(const Dev& d, int center, int iter, int phase, const float* cost_array /*[8][MV]*/,
	const float* priors, uint8_t* vw /*[MV], zeroed*/, uint32_t* sel_mask, float* weight_norm) {
	const int S = d.params.num_images - 1;
	float probs[MV];
	const float thr = (float)(0.8 * dvp_expf((iter) * (iter) / (-90.0f)));
	for (int i = 0; i < S; i++) {
		float count = 0;
		int count_false = 0;
		float tmpw = 0;
		for (int j = 0; j < 8; j++) {
			const float cst = cost_array[j * MV + i];
			if (cst < thr) { tmpw += dvp_expf(cst * cst / (-0.18f)); count++; }
			if (cst > 1.2f) count_false++;
		}
		float pr = 0.0f;
		if (count > 2 && count_false < 3) pr = tmpw / count;
		else if (count_false < 3) pr = dvp_expf(thr * thr / (-0.32f));
		probs[i] = pr * priors[i];
	}
	// TransformPDFToCDF (APD.cu:356-370)
	float psum = 0.0f;
	for (int i = 0; i < S; ++i) psum += probs[i];
	const float inv = 1.0f / psum;
	float cum = 0.0f;
	for (int i = 0; i < S; ++i) { cum += probs[i] * inv; probs[i] = cum; }
	Rng rv(d.seed, (uint32_t)center, rng_site(phase, iter, SUB_VIEW));
	for (int s = 0; s < 15; ++s) {
		const float rp = rv.uniform() - FLT_EPSILON;
		for (int v = 0; v < S; ++v)
			if (probs[v] > rp) { vw[v] += 1; break; }
	}
	uint32_t m = 0;
	float wn = 0;
	for (int i = 0; i < S; ++i)
		if (vw[i] > 0) { set_bit(&m, i); wn += vw[i]; }
	*sel_mask = m;
	*weight_norm = wn;
}

// edge-adaptive sample search of one direction (APD.cu:2047-2081 / 2104-2118).
// pass 0: adaptive step; pass 1: 11 samples at stride 2.  Returns the position or -1.
DVP_HD int strong_sample_search(const Dev& d, int px, int py, int k, int pass) {
	const int W = d.width, H = d.height;
	const int center = py * W + px;
	const int dxs[8] = { 0, 0, -1, 1, -1, 1, -1, 1 };
	const int dys[8] = { -1, 1, 0, 0, -1, 1, 1, -1 };
	const int dx = dxs[k], dy = dys[k];
	int step_num = 11, step_len = 2;
	if (pass == 0) {
		const float max_edge_dist = DVP_MAX(H, W) / 30.0f;
		const s2 ep = d.edge_neigh[(size_t)center * 8 + k];
		const double ex = (double)(ep.x - px), ey = (double)(ep.y - py);
		float dist = (float)sqrt(ex * ex + ey * ey);
		if (k >= 4) dist = (float)((double)dist / sqrt(2.0));
		if (d.edge[center]) {
			dist = 22.0f;
		} else if (ep.y == -1 || dist >= max_edge_dist) {   // `!edge_pt.x == -1` is always false (APD.cu:2059)
			dist = max_edge_dist;
			if (k >= 4) dist = (float)((double)dist / sqrt(2.0));
		}
		step_num = DVP_MIN(DVP_MAX(11, (int)(1.0f * dist / 2)), 22);
		step_len = DVP_MAX((int)(1.0f * dist / step_num), 2);
		if (k < 4 && step_len % 2 == 1) step_len -= 1;
	}
	int fx = 0, fy = 0;
	if (k > 4) { if (k % 2) fx = dx; else fy = dy; }
	int best = -1;
	float min_cost = FLT_MAX;
	// step_num <= 22: fixed trip count with predicated loads so that all samples are in flight at
	// once instead of one dependent L2 round trip per sample
	float cs[22];
	int pcs[22];
#pragma unroll
	for (int step = 0; step < 22; ++step) {
		const int tx = px + 5 * dx + step * step_len * dx + fx;
		const int ty = py + 5 * dy + step * step_len * dy + fy;
		const bool ok = step < step_num && tx >= 0 && ty >= 0 && tx < W && ty < H;
		pcs[step] = ok ? tx + ty * W : -1;
		cs[step] = ok ? d.costs_snap[tx + ty * W] : 0.0f;
	}
#pragma unroll
	for (int step = 0; step < 22; ++step) {
		if (pcs[step] >= 0 && min_cost > cs[step]) { best = pcs[step]; min_cost = cs[step]; }
	}
	return (min_cost < FLT_MAX) ? best : -1;
}

// The sample positions of the 16 propagation slots of a pixel (8 edge-adaptive + 8 fixed, -1 = none)
// depend only on the pre-launch snapshot and the edge priors, so they are found by a separate,
// light launch (dozens of waves per SIMD hide the two dependent round trips per slot) and the
// register-starved update kernel reads one word per slot.
DVP_HD void strong_search_px(const Dev& d, int px, int py) {
	const int center = py * d.width + px;
	if (d.weak_info[center] == DVP_WEAK) return;
	const size_t L = (size_t)d.width * d.height;
	const bool is_edge = d.edge[center] != 0;
	for (int slot = 0; slot < 16; ++slot) {
		int pos = -1;
		if (slot < 8) pos = strong_sample_search(d, px, py, slot, 0);
		else if (!is_edge) pos = strong_sample_search(d, px, py, slot - 8, 1);
		d.search_pos[(size_t)slot * L + center] = pos;
	}
}

// CheckerboardPropagationStrong + PlaneHypothesisRefinementStrong
// (APD.cu:2010-2141, 2462-2567, 2725-2737, 1311-1383), use_edge branch.
//
// The 8 + 8 propagation candidates, the current plane and the 6 refinement hypotheses are walked
// by ONE loop with a single inlined copy of the 36-tap evaluation (23 "slots"); per slot a
// prologue picks the plane and the views to evaluate, an epilogue consumes the cost vector.
// After view selection only views with non-zero weight are evaluated: the reference evaluates all
// S and multiplies the others by a zero weight, which is the same value.
// MV = capacity of the per-view private arrays (cost_array[8][MV], cv, priors, vw): 32 covers
// every legal view count; the engine launches the MV = 8 instantiation when S <= 8, which keeps
// 1.2 KB/lane of zero-filled scratch out of the cache hierarchy (same values either way).
template <int SMP, int MV = 32>
DVP_HD void strong_update_px(const Dev& d, int px, int py, PatchTab tab, int iter, unsigned long long* nevals) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const uint32_t all_views = (S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);

	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
	}
	float cost_array[8 * MV];
	for (int i = 0; i < 8 * MV; ++i) cost_array[i] = 0.0f;
	cost_array[0] = 2.0f;   // `= { 2.0f }` sets one element (APD.cu:2032)
	uint32_t flag = 0;      // bit k: direction k has a sample
	int positions[8];
	for (int k = 0; k < 8; ++k) positions[k] = 0;
	const bool is_edge = d.edge[center] != 0;
	const float good_thr = 0.8f * dvp_expf((iter) * (iter) / (-90.0f));

	uint8_t vw[MV];
	for (int i = 0; i < MV; ++i) vw[i] = 0;
	uint32_t sel_mask = 0;
	float weight_norm = 0.0f;
	float final_costs[8];
	int min_cost_idx = 0;
	float cost_now = 0.0f, costs_center = 0.0f, depth_now = 0.0f;
	f4 plane_now = mk4(0, 0, 0, 0);
	bool selected_views_written = false;
	// refinement hypotheses (APD.cu:1359-1360)
	float ref_depths[6];
	f4 ref_normals[6];

	float cv[MV];
	for (int slot = 0; slot < 23; ++slot) {
		// hypotheses 3 and 4 of the refinement are the same plane (both GeneratePerturbedNormal calls
		// return the input normal, APD.cu:1354-1360): the second one can never be accepted after the
		// first was tested (same cost, strict `<`), so it is not evaluated.
		if (slot == 21) continue;
		// ---- prologue: which plane, which views ----
		f4 plane = mk4(0, 0, 1, 1);
		uint32_t mask = 0;
		int pos = -1;
		if (slot < 16) {
			pos = d.search_pos[(size_t)slot * ((size_t)W * d.height) + center];   // strong_search_px (-1 on edge pixels for slots 8-15)
			if (pos >= 0) { plane = d.planes_snap[pos]; mask = all_views; }
		} else if (slot == 16) {
			// view selection (APD.cu:2462-2530)
			float priors[MV];
			for (int i = 0; i < MV; ++i) priors[i] = 0.0f;
			const int nb[4] = { center - W, center + W, center - 1, center + 1 };
			for (int i = 0; i < 4; ++i) {
				if ((flag >> (2 * i)) & 1) {   // guards flag[0],[2],[4],[6] (APD.cu:2471)
					const uint32_t sv = d.selected_views[nb[i]];
					for (int j = 0; j < S; ++j) priors[j] += is_set(sv, j) ? 0.9f : 0.1f;
				}
			}
			joint_view_selection<MV>(d, center, iter, PH_STRONG, cost_array, priors, vw, &sel_mask, &weight_norm);
			uint8_t* gvw = d.view_weight + (size_t)center * 32;
			for (int i = 0; i < 32; ++i) gvw[i] = i < MV ? vw[i < MV ? i : 0] : (uint8_t)0;
			for (int k = 0; k < 8; ++k) {
				float fc = 0.0f;
				for (int j = 0; j < S; ++j)
					if (vw[j] > 0) fc += vw[j] * cost_array[k * MV + j];
				final_costs[k] = fc / weight_norm;
			}
			min_cost_idx = 0;   // FindMinCostIndex (ties -> last, APD.cu:155-166)
			{
				float mc = final_costs[0];
				for (int k = 1; k < 8; ++k)
					if (final_costs[k] <= mc) { mc = final_costs[k]; min_cost_idx = k; }
			}
			plane = d.planes_snap[center];
			mask = sel_mask;
		} else {
			const int i = slot - 17;
			plane = ref_normals[i];
			plane.w = distance_to_origin(rc, px, py, ref_depths[i], plane);
			mask = sel_mask;
		}

		// ---- evaluate ----
		if (mask) {
			for (int v = 0; v < S; ++v) {
				if ((mask >> v) & 1) {
					cv[v] = ncc_old<SMP>(d, c, px, py, v + 1, plane);
					if (nevals) *nevals += 1;
				}
			}
		}

		// ---- epilogue ----
		if (slot < 8) {
			if (pos >= 0) {
				flag |= 1u << slot;
				positions[slot] = pos;
				for (int v = 0; v < S; ++v) cost_array[slot * MV + v] = cv[v];
			}
		} else if (slot < 16) {
			const int k = slot - 8;
			if (pos >= 0) {
				const bool had = (flag >> k) & 1;
				flag |= 1u << k;
				int good0 = 0, good1 = 0, bad0 = 0, bad1 = 0;
				for (int j = 0; j < S; ++j) {
					const float a = cost_array[k * MV + j], b = cv[j];
					if (a < good_thr) good0++;
					if (a > 1.2f) bad0++;
					if (b < good_thr) good1++;
					if (b > 1.2f) bad1++;
				}
				if (!had || good1 > good0 || (good1 == good0 && bad1 < bad0)) {
					positions[k] = pos;
					for (int j = 0; j < S; ++j) cost_array[k * MV + j] = cv[j];
				}
			}
		} else if (slot == 16) {
			// cost of the current plane under the new weights, adoption of the best neighbour
			// (APD.cu:2546-2567); a zero weight_norm makes everything NaN and every `<` false.
			float cn = 0.0f;
			for (int v = 0; v < S; ++v)
				if (vw[v] > 0) cn += vw[v] * cv[v];
			cost_now = cn / weight_norm;
			costs_center = cost_now;
			plane_now = d.planes_snap[center];
			depth_now = depth_from_plane(rc, plane_now, px, py);
			if ((flag >> min_cost_idx) & 1) {
				const f4 cand = d.planes_snap[positions[min_cost_idx]];
				const float db = depth_from_plane(rc, cand, px, py);
				if (db >= P.depth_min && db <= P.depth_max && final_costs[min_cost_idx] < cost_now) {
					depth_now = db;
					plane_now = cand;
					cost_now = final_costs[min_cost_idx];
					selected_views_written = true;
				}
			}
			// refinement hypotheses from the values at entry (APD.cu:1333-1360)
			Rng rd(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_DEPTH_RAND));
			Rng rn(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_NORMAL));
			Rng rp(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_DEPTH_PERT));
			const float depth_rand = rd.uniform() * (P.depth_max - P.depth_min) + P.depth_min;
			if (selected_views_written) d.selected_views[center] = sel_mask;   // read by random_normal_yzl
			const f4 n_rand = random_normal_yzl(d, px, py, rn, depth_now);
			const float dmin_p = (1 - 0.02f) * depth_now, dmax_p = (1 + 0.02f) * depth_now;
			const float depth_pert = rp.uniform() * (dmax_p - dmin_p) + dmin_p;
			f4 n_pert = plane_now;   // GeneratePerturbedNormal returns the normalised input (APD.cu:617-661)
			normalize3(&n_pert);
			ref_depths[0] = depth_rand; ref_normals[0] = plane_now;
			ref_depths[1] = depth_now;  ref_normals[1] = n_rand;
			ref_depths[2] = depth_rand; ref_normals[2] = n_rand;
			ref_depths[3] = depth_now;  ref_normals[3] = n_pert;
			ref_depths[4] = depth_now;  ref_normals[4] = n_pert;
			ref_depths[5] = depth_pert; ref_normals[5] = plane_now;
		} else {
			float tc = 0.0f;
			for (int j = 0; j < S; ++j)
				if (vw[j] > 0) tc += vw[j] * cv[j];
			tc /= weight_norm;
			const float db = depth_from_plane(rc, plane, px, py);
			if (db >= P.depth_min && db <= P.depth_max && tc < cost_now) {
				depth_now = db;
				plane_now = plane;
				cost_now = tc;
			}
		}
	}

	if (P.state == DVP_REFINE_INIT) {
		if (cost_now < costs_center - 0.1) {   // double comparison (APD.cu:2728)
			costs_center = cost_now;
			d.planes[center] = plane_now;
		}
	} else {
		costs_center = cost_now;
		d.planes[center] = plane_now;
	}
	d.costs[center] = costs_center;
}

// ===== split form of the strong update (engine launches only; same evaluations, same operations, same bits) ==========
// Measured on MI355X: the 36-tap evaluator alone runs at 32.7 G evaluations/s, inside the 23-slot loop of
// strong_update_px at 20-22, and the loop's decision logic alone (scratch-resident cost arrays, view selection, random
// normals at two waves per SIMD) costs 28 of the kernel's 109 ms.  The update is therefore issued as three launches:
//   dvp_strong_eval    strong_eval_px    the 16 propagation slots + the current plane, all S views each: pure functions of
//                                        the pre-launch snapshot; nothing live but the evaluator -> Dev::slot_costs
//   dvp_strong_decide  strong_decide_px  slot bookkeeping, view selection, adoption, refinement hypotheses: no evaluator,
//                                        no patch table, cost vectors in registers -> Dev::strong_rec (48 B per pixel)
//   dvp_strong_refine  strong_refine_px  the five refinement hypotheses against the selected views, acceptance, write-back
// strong_update_px stays the definition (host emulation, S > 16, dvp_run_stage A/B with DVP_STRONG_SPLIT=0).
constexpr int kSlotCur = 16;                 // slot_costs slot of the pixel's current plane
constexpr int kSlotCount = 17;
enum { SR_PLANE = 0, SR_DEPTH = 4, SR_COST = 5, SR_CENTER = 6, SR_DRAND = 7, SR_DPERT = 8, SR_NRAND = 9, SR_DUP = 12 /* 3 words: strong_slot_sources */, SR_FIELDS = 15 };
DVP_HD size_t half_index(const Dev& d, int px, int py) { return (size_t)py * d.half_w + (size_t)(px >> 1); }
// Layout of Dev::slot_costs.  Round 6: [pixel of the colour][slot][view] — a pixel's 17 x S costs are one contiguous record.
// The evaluation launch's lanes are (pixel, slot) items, pixel-major: neighbouring lanes now write neighbouring 4 S-byte
// vectors (rounds 3-5, [slot][view][pixel]: 64 lines per store instruction), and a lane of the decision launch finds its 17 x S
// costs in five cache lines instead of 153 (strong update 466 -> 455 ms per cfg3 pass).
#ifndef DVP_SLOT_LAYOUT
#define DVP_SLOT_LAYOUT 1   // 0: [slot][view][pixel] (A/B)
#endif
// place of a slot's vector inside the pixel's record: the decision step takes the slots in pairs (k, 8 + k) — the adaptive and the
// fixed-stride sample of direction k — so the pairs lie side by side and a lane walks its record front to back, every cache line once
DVP_HD int slot_place(int slot) {
#if DVP_SLOT_LAYOUT
	return slot < 8 ? 2 * slot : (slot < 16 ? 2 * (slot - 8) + 1 : slot);
#else
	return slot;
#endif
}
DVP_HD size_t slot_cost_index(const Dev& d, int slot, int v, int px, int py) {
	const size_t Lh = (size_t)d.half_w * (size_t)d.height;
	const int S = d.params.num_images - 1;
#if DVP_SLOT_LAYOUT
	return (half_index(d, px, py) * kSlotCount + (size_t)slot_place(slot)) * (size_t)S + (size_t)v;
#else
	return (size_t)(slot * S + v) * Lh + half_index(d, px, py);
#endif
}
// where the decision step finds cost(slot, view) of its pixel: base[slot * slot_stride + view * view_stride]
struct SlotCostView { const float* base; size_t slot_stride, view_stride; };
DVP_HD SlotCostView slot_cost_view(const Dev& d, int px, int py) {
	const size_t Lh = (size_t)d.half_w * (size_t)d.height;
	const int S = d.params.num_images - 1;
	SlotCostView r;
#if DVP_SLOT_LAYOUT
	r.base = d.slot_costs + half_index(d, px, py) * kSlotCount * (size_t)S;
	r.slot_stride = (size_t)S;
	r.view_stride = 1;
#else
	r.base = d.slot_costs + half_index(d, px, py);
	r.slot_stride = (size_t)S * Lh;
	r.view_stride = Lh;
#endif
	return r;
}
// Round 4: the cost vector of a slot is a pure function of (pixel, plane), and the 17 planes of a pixel repeat — the
// edge-adaptive and the fixed search of a direction often end on the same pixel, and propagation itself makes neighbours
// carry bit-identical planes (11 % of the launch site's evaluations at cfg3).  Every bitwise-DISTINCT plane is evaluated
// once; a slot whose plane an earlier slot has already had is served by that slot's vector (same inputs through the same
// code: the same bits): src[slot] = the earliest slot with the same plane (itself when it is the first), 4 bits per slot in
// w[0] (slots 0-7) and w[1] (8-15), w[2] = src[16].  Returns the mask of the slots to evaluate.
DVP_HD bool same_plane_bits(const f4 a, const f4 b) {
	return f32_bits(a.x) == f32_bits(b.x) && f32_bits(a.y) == f32_bits(b.y) && f32_bits(a.z) == f32_bits(b.z) && f32_bits(a.w) == f32_bits(b.w);
}
DVP_HD uint32_t strong_slot_sources(const Dev& d, int center, uint32_t w[3]) {
	const size_t L = (size_t)d.width * d.height;
	f4 pl[kSlotCount];
	uint32_t have = 0, uniq = 0;
#pragma unroll
	for (int slot = 0; slot < kSlotCount; ++slot) {
		const int pos = slot < 16 ? d.search_pos[(size_t)slot * L + center] : center;
		pl[slot] = mk4(0.0f, 0.0f, 0.0f, 0.0f);
		if (pos >= 0) { pl[slot] = d.planes_snap[pos]; have |= 1u << slot; }
	}
	w[0] = w[1] = w[2] = 0;
#pragma unroll
	for (int slot = 0; slot < kSlotCount; ++slot) {
		int src = slot;
#pragma unroll
		for (int u = slot - 1; u >= 0; --u)     // ends with the EARLIEST equal slot, which is an evaluated one
			if (((have >> u) & 1u) && same_plane_bits(pl[u], pl[slot])) src = u;
		if (!((have >> slot) & 1u)) continue;
		if (src == slot) uniq |= 1u << slot;
		if (slot < 8) w[0] |= (uint32_t)src << (4 * slot);
		else if (slot < 16) w[1] |= (uint32_t)src << (4 * (slot - 8));
		else w[2] = (uint32_t)src;
	}
	return uniq;
}
DVP_HD int strong_slot_source(uint32_t w0, uint32_t w1, uint32_t w2, int slot) {
	return slot < 8 ? (int)((w0 >> (4 * slot)) & 15u) : slot < 16 ? (int)((w1 >> (4 * (slot - 8))) & 15u) : (int)w2;
}
// one (pixel, slot) item: the slot's plane against all S views -> slot_costs
template <int SMP>
DVP_HD void strong_eval_item(const Dev& d, const PatchCtx& c, int px, int py, int slot, bool store, unsigned long long* nevals) {
	const int W = d.width;
	const int center = py * W + px;
	const int S = d.params.num_images - 1;
	const size_t L = (size_t)W * d.height;
	const int pos = slot < 16 ? d.search_pos[(size_t)slot * L + center] : center;
	const f4 plane = d.planes_snap[pos];
	for (int v = 0; v < S; ++v) {
		const float cost = ncc_old<SMP>(d, c, px, py, v + 1, plane);
		if (store) DVP_NT_STORE(4, &d.slot_costs[slot_cost_index(d, slot, v, px, py)], cost);
	}
	if (nevals && store) *nevals += (unsigned long long)S;
}
// the pixel's share of the evaluation launch, one lane per pixel (the definition: host emulation, DVP_EVAL_ITEMS=0; the
// engine's default is the same items compacted over the lanes of a wave, dvp_strong_eval_items)
template <int SMP>
DVP_HD void strong_eval_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int W = d.width;
	const int center = py * W + px;
	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
	}
	uint32_t w[3];
	const uint32_t uniq = strong_slot_sources(d, center, w);
	const size_t Lh = (size_t)d.half_w * (size_t)d.height, hi = half_index(d, px, py);
	uint32_t* dup = reinterpret_cast<uint32_t*>(d.strong_rec) + hi;
	dup[SR_DUP * Lh] = w[0]; dup[(SR_DUP + 1) * Lh] = w[1]; dup[(SR_DUP + 2) * Lh] = w[2];
	for (uint32_t m = uniq; m; m &= m - 1) strong_eval_item<SMP>(d, c, px, py, dvp_ctz(m), true, nevals);
}

// the S (<= MV) costs of one slot.  Pixel-major records hold them contiguously: 16-byte loads (4-byte aligned; the pieces beyond
// S belong to the next slot — the buffer carries 64 bytes of slack after the last record — and are not used)
template <int MV>
DVP_HD void load_slot_costs(const float* sc, size_t view_stride, int S, float* out) {
#if defined(__HIP_DEVICE_COMPILE__) && DVP_SLOT_LAYOUT
	typedef float f4v __attribute__((ext_vector_type(4), aligned(4)));
	constexpr int Q = (MV + 3) / 4;
	f4v q[Q];
#pragma unroll
	for (int i = 0; i < Q; ++i) q[i] = DVP_NT_LOAD(8, &reinterpret_cast<const f4v*>(sc)[i]);
#pragma unroll
	for (int v = 0; v < MV; ++v) out[v] = v < S ? q[v >> 2][v & 3] : 0.0f;
#else
#pragma unroll
	for (int v = 0; v < MV; ++v) out[v] = v < S ? sc[(size_t)v * view_stride] : 0.0f;
#endif
}

// Everything of strong_update_px between the propagation evaluations and the refinement evaluations, statement for
// statement, with the cost vectors in registers (all loops over directions / views are unrolled; MV >= S).
template <int MV>
DVP_HD void strong_decide_px(const Dev& d, int px, int py, int iter) {
	const SlotCostView cv = slot_cost_view(d, px, py);
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const size_t L = (size_t)W * d.height;
	const size_t Lh = (size_t)d.half_w * (size_t)d.height, hi = half_index(d, px, py);
	const float good_thr = 0.8f * dvp_expf((iter) * (iter) / (-90.0f));
	const uint32_t sel_entry = d.selected_views[center];   // (the pixel's own word: no other pixel of this colour writes it)
	float ca[8][MV];
	uint32_t flag = 0;
	int positions[8];
	// which slot's vector serves a slot (strong_slot_sources, written by the evaluation launch)
	const uint32_t* dup = reinterpret_cast<const uint32_t*>(d.strong_rec) + hi;
	const uint32_t dw0 = dup[SR_DUP * Lh], dw1 = dup[(SR_DUP + 1) * Lh], dw2 = dup[(SR_DUP + 2) * Lh];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		positions[k] = 0;
#pragma unroll
		for (int v = 0; v < MV; ++v) ca[k][v] = 0.0f;
	}
	ca[0][0] = 2.0f;   // `= { 2.0f }` sets one element (APD.cu:2032)
	// The reference walks slots 0-7 (APD.cu:2047-2090) and then slots 8-15 (APD.cu:2104-2137); direction k of the second walk only
	// looks at what direction k of the first left, so the two are taken together here, direction by direction
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		{
			const int pos = d.search_pos[(size_t)k * L + center];
			if (pos >= 0) {
				flag |= 1u << k;
				positions[k] = pos;
				load_slot_costs<MV>(cv.base + (size_t)slot_place(strong_slot_source(dw0, dw1, dw2, k)) * cv.slot_stride, cv.view_stride, S, ca[k]);
			}
		}
		// slot 8 + k: the fixed-stride sample replaces the adaptive one if it is better
		const int pos = d.search_pos[(size_t)(8 + k) * L + center];
		if (pos >= 0) {
			const bool had = (flag >> k) & 1;
			flag |= 1u << k;
			float cb[MV];
			int good0 = 0, good1 = 0, bad0 = 0, bad1 = 0;
			load_slot_costs<MV>(cv.base + (size_t)slot_place(strong_slot_source(dw0, dw1, dw2, 8 + k)) * cv.slot_stride, cv.view_stride, S, cb);
#pragma unroll
			for (int j = 0; j < MV; ++j) {
				if (j < S) {
					const float a = ca[k][j], b = cb[j];
					if (a < good_thr) good0++;
					if (a > 1.2f) bad0++;
					if (b < good_thr) good1++;
					if (b > 1.2f) bad1++;
				}
			}
			if (!had || good1 > good0 || (good1 == good0 && bad1 < bad0)) {
				positions[k] = pos;
#pragma unroll
				for (int j = 0; j < MV; ++j)
					if (j < S) ca[k][j] = cb[j];
			}
		}
	}
	// ---- view selection (APD.cu:2462-2530), joint_view_selection on the register arrays ----
	float priors[MV];
#pragma unroll
	for (int i = 0; i < MV; ++i) priors[i] = 0.0f;
	{
		const int nb[4] = { center - W, center + W, center - 1, center + 1 };
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if ((flag >> (2 * i)) & 1) {   // guards flag[0],[2],[4],[6] (APD.cu:2471)
				const uint32_t sv = d.selected_views[nb[i]];
#pragma unroll
				for (int j = 0; j < MV; ++j)
					if (j < S) priors[j] += is_set(sv, j) ? 0.9f : 0.1f;
			}
		}
	}
	float probs[MV];
	const float thr = (float)(0.8 * dvp_expf((iter) * (iter) / (-90.0f)));
#pragma unroll
	for (int i = 0; i < MV; ++i) {
		probs[i] = 0.0f;
		if (i < S) {
			float count = 0;
			int count_false = 0;
			float tmpw = 0;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const float cst = ca[j][i];
				if (cst < thr) { tmpw += dvp_expf(cst * cst / (-0.18f)); count++; }
				if (cst > 1.2f) count_false++;
			}
			float pr = 0.0f;
			if (count > 2 && count_false < 3) pr = tmpw / count;
			else if (count_false < 3) pr = dvp_expf(thr * thr / (-0.32f));
			probs[i] = pr * priors[i];
		}
	}
	float psum = 0.0f;
#pragma unroll
	for (int i = 0; i < MV; ++i)
		if (i < S) psum += probs[i];
	const float inv = 1.0f / psum;
	float cum = 0.0f;
#pragma unroll
	for (int i = 0; i < MV; ++i)
		if (i < S) { cum += probs[i] * inv; probs[i] = cum; }
	int vw[MV];
#pragma unroll
	for (int i = 0; i < MV; ++i) vw[i] = 0;
	Rng rv(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_VIEW));
	for (int s = 0; s < 15; ++s) {
		const float rp = rv.uniform() - FLT_EPSILON;
		bool done = false;
#pragma unroll
		for (int v = 0; v < MV; ++v)
			if (v < S && !done && probs[v] > rp) { vw[v] += 1; done = true; }
	}
	uint32_t sel_mask = 0;
	float weight_norm = 0;
#pragma unroll
	for (int i = 0; i < MV; ++i)
		if (i < S && vw[i] > 0) { set_bit(&sel_mask, i); weight_norm += vw[i]; }
	{
		uint8_t* gvw = d.view_weight + (size_t)center * 32;
		uint32_t words[8];
#pragma unroll
		for (int w8 = 0; w8 < 8; ++w8) {
			uint32_t x = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b)
				if (w8 * 4 + b < MV) x |= ((uint32_t)vw[w8 * 4 + b] & 255u) << (8 * b);
			words[w8] = x;
		}
		uint32_t* g32 = reinterpret_cast<uint32_t*>(gvw);   // 32-byte records: aligned
#pragma unroll
		for (int w8 = 0; w8 < 8; ++w8) g32[w8] = words[w8];
	}
	float final_costs[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		float fc = 0.0f;
#pragma unroll
		for (int j = 0; j < MV; ++j)
			if (j < S && vw[j] > 0) fc += vw[j] * ca[k][j];
		final_costs[k] = fc / weight_norm;
	}
	int min_cost_idx = 0;   // FindMinCostIndex (ties -> last, APD.cu:155-166)
	{
		float mc = final_costs[0];
#pragma unroll
		for (int k = 1; k < 8; ++k)
			if (final_costs[k] <= mc) { mc = final_costs[k]; min_cost_idx = k; }
	}
	// ---- cost of the current plane under the new weights, adoption of the best neighbour (APD.cu:2546-2567) ----
	float cn = 0.0f;
#pragma unroll
	for (int v = 0; v < MV; ++v)
		if (v < S && vw[v] > 0) cn += vw[v] * cv.base[(size_t)slot_place((int)dw2) * cv.slot_stride + (size_t)v * cv.view_stride];
	float cost_now = cn / weight_norm;
	const float costs_center = cost_now;
	f4 plane_now = d.planes_snap[center];
	float depth_now = depth_from_plane(rc, plane_now, px, py);
	bool selected_views_written = false;
	{
		float fmin = final_costs[0];
		int pmin = positions[0];
#pragma unroll
		for (int k = 1; k < 8; ++k)
			if (k == min_cost_idx) { fmin = final_costs[k]; pmin = positions[k]; }
		if ((flag >> min_cost_idx) & 1) {
			const f4 cand = d.planes_snap[pmin];
			const float db = depth_from_plane(rc, cand, px, py);
			if (db >= P.depth_min && db <= P.depth_max && fmin < cost_now) {
				depth_now = db;
				plane_now = cand;
				cost_now = fmin;
				selected_views_written = true;
			}
		}
	}
	// ---- refinement hypotheses from the values at entry (APD.cu:1333-1360) ----
	Rng rd(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_DEPTH_RAND));
	Rng rn(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_NORMAL));
	Rng rp(d.seed, (uint32_t)center, rng_site(PH_STRONG, iter, SUB_DEPTH_PERT));
	const float depth_rand = rd.uniform() * (P.depth_max - P.depth_min) + P.depth_min;
	if (selected_views_written) d.selected_views[center] = sel_mask;   // (what random_normal_yzl reads)
#if defined(DVP_DECIDE_YZL_GENERIC)   // A/B: the per-view walk with its scratch array
	const f4 n_rand = random_normal_yzl(d, px, py, rn, depth_now);
#else
	const f4 n_rand = MV <= 16 ? random_normal_yzl_views<(MV <= 16 ? MV : 16)>(d, px, py, rn, depth_now, selected_views_written ? sel_mask : sel_entry)
	                           : random_normal_yzl(d, px, py, rn, depth_now);
#endif
	const float dmin_p = (1 - 0.02f) * depth_now, dmax_p = (1 + 0.02f) * depth_now;
	const float depth_pert = rp.uniform() * (dmax_p - dmin_p) + dmin_p;
	float* rec = d.strong_rec + hi;
	rec[(SR_PLANE + 0) * Lh] = plane_now.x; rec[(SR_PLANE + 1) * Lh] = plane_now.y; rec[(SR_PLANE + 2) * Lh] = plane_now.z; rec[(SR_PLANE + 3) * Lh] = plane_now.w;
	rec[SR_DEPTH * Lh] = depth_now; rec[SR_COST * Lh] = cost_now; rec[SR_CENTER * Lh] = costs_center;
	rec[SR_DRAND * Lh] = depth_rand; rec[SR_DPERT * Lh] = depth_pert;
	rec[(SR_NRAND + 0) * Lh] = n_rand.x; rec[(SR_NRAND + 1) * Lh] = n_rand.y; rec[(SR_NRAND + 2) * Lh] = n_rand.z;
}

// PlaneHypothesisRefinementStrong's evaluations and acceptance (APD.cu:1361-1383) + the write-back (APD.cu:2725-2737)
template <int SMP, bool LANE_WALK = false>
DVP_HD void strong_refine_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const size_t Lh = (size_t)d.half_w * (size_t)d.height, hi = half_index(d, px, py);
	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
	}
	const float* rec = d.strong_rec + hi;
	f4 plane_now = mk4(rec[(SR_PLANE + 0) * Lh], rec[(SR_PLANE + 1) * Lh], rec[(SR_PLANE + 2) * Lh], rec[(SR_PLANE + 3) * Lh]);
	float depth_now = rec[SR_DEPTH * Lh], cost_now = rec[SR_COST * Lh], costs_center = rec[SR_CENTER * Lh];
	const float depth_rand = rec[SR_DRAND * Lh], depth_pert = rec[SR_DPERT * Lh];
	const f4 n_rand = mk4(rec[(SR_NRAND + 0) * Lh], rec[(SR_NRAND + 1) * Lh], rec[(SR_NRAND + 2) * Lh], 0.0f);
	f4 n_pert = plane_now;   // GeneratePerturbedNormal returns the normalised input (APD.cu:617-661)
	normalize3(&n_pert);
	// view weights of this launch (strong_decide_px wrote them): two 16-byte loads
	const uint32_t* g32 = reinterpret_cast<const uint32_t*>(d.view_weight + (size_t)center * 32);
	uint32_t wq[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) wq[i] = g32[i];
	float weight_norm = 0;
	for (int i = 0; i < S; ++i) {
		const int wv = (int)((wq[i >> 2] >> (8 * (i & 3))) & 255u);
		if (wv > 0) weight_norm += wv;
	}
	// hypotheses 3 and 4 are the same plane (strong_update_px): the second one is not evaluated
	const float hyp_depth[5] = { depth_rand, depth_now, depth_rand, depth_now, depth_pert };
	const f4 hyp_normal[5] = { plane_now, n_rand, n_rand, n_pert, plane_now };
	const float depth_entry = depth_now;
	const f4 plane_entry = plane_now;
	(void)depth_entry; (void)plane_entry;
	// A hypothesis is adopted iff its depth is in range and its weighted cost is below the running best (strict).
	// Two exact short cuts (the cost itself is only kept when adopted): (i) out of range -> not evaluated; (ii) the
	// weighted sum only grows — every term is a weight > 0 times a cost in [0, 2], IEEE addition and division are
	// monotone — so once the partial sum / weight_norm is not below cost_now the final one cannot be: the remaining
	// views are not evaluated.  Random hypotheses are rejected after one or two views instead of all selected ones.
	if (LANE_WALK) {
		// Round 4: every lane walks its OWN (hypothesis, view) sequence — one evaluation per trip of the loop below for
		// every lane that still has one — instead of the wave stepping through hypotheses and views together (where a
		// lane waits while any neighbour evaluates a view it did not select or a hypothesis it has already rejected:
		// lane utilisation 0.47).  Same evaluations per pixel in the same order, hence the same bits; the view index is
		// a per-lane value (ncc_old_lane).
		uint32_t sel = 0;     // views with a weight > 0
		for (int j = 0; j < S; ++j)
			if (((wq[j >> 2] >> (8 * (j & 3))) & 255u) > 0) sel |= 1u << j;
		int hyp = 0;
		bool busy = false;
		f4 plane = plane_now;
		float db = 0.0f, tc = 0.0f;
		uint32_t todo = 0;    // views of the current hypothesis not yet evaluated
		for (;;) {
			while (!busy && hyp < 5) {
				plane = hyp_normal[hyp];
				plane.w = distance_to_origin(rc, px, py, hyp_depth[hyp], plane);
				db = depth_from_plane(rc, plane, px, py);
				++hyp;
				if (!(db >= P.depth_min && db <= P.depth_max)) continue;
				if (!(weight_norm > 0.0f)) continue;      // no selected view: 0 / 0 = NaN, never below cost_now
				tc = 0.0f;
				todo = sel;
				busy = true;
			}
			if (!busy) break;
			const int j = dvp_ctz(todo);
			todo &= todo - 1;
			uint32_t q = wq[0];
#pragma unroll
			for (int k = 1; k < 8; ++k) q = (j >> 2) == k ? wq[k] : q;
			const int wv = (int)((q >> (8 * (j & 3))) & 255u);
			tc += wv * ncc_old_lane<SMP>(d, c, px, py, j + 1, plane);
			if (nevals) *nevals += 1;
			const bool alive = tc / weight_norm < cost_now;
			if (!alive) busy = false;
			else if (todo == 0) {      // the complete sum is below the running best
				depth_now = db;
				plane_now = plane;
				cost_now = tc / weight_norm;
				busy = false;
			}
		}
	} else
	for (int i = 0; i < 5; ++i) {
		f4 plane = hyp_normal[i];
		plane.w = distance_to_origin(rc, px, py, hyp_depth[i], plane);
		const float db = depth_from_plane(rc, plane, px, py);
		if (!(db >= P.depth_min && db <= P.depth_max)) continue;
		float tc = 0.0f;
		bool alive = weight_norm > 0.0f;     // no selected view: 0 / 0 = NaN, never below cost_now
		for (int j = 0; j < S && alive; ++j) {
			const int wv = (int)((wq[j >> 2] >> (8 * (j & 3))) & 255u);
			if (wv > 0) {
				tc += wv * ncc_old<SMP>(d, c, px, py, j + 1, plane);
				if (nevals) *nevals += 1;
				alive = tc / weight_norm < cost_now;
			}
		}
		if (alive) {      // == (tc / weight_norm < cost_now) of the complete sum
			depth_now = db;
			plane_now = plane;
			cost_now = tc / weight_norm;
		}
	}
	if (P.state == DVP_REFINE_INIT) {
		if (cost_now < costs_center - 0.1) {   // double comparison (APD.cu:2728)
			costs_center = cost_now;
			d.planes[center] = plane_now;
		}
	} else {
		costs_center = cost_now;
		d.planes[center] = plane_now;
	}
	d.costs[center] = costs_center;
}

// GetDepthandNormal (APD.cu:3167-3182)
DVP_HD void get_depth_normal_px(const Dev& d, int px, int py) {
	const int center = py * d.width + px;
	f4 pl = d.planes[center];
	pl.w = depth_from_plane(d.cameras[0], pl, px, py);
	d.planes[center] = normal_cam_to_world(d.cameras[0], pl);
}

// CheckerboardFilterStrong (APD.cu:3184-3294): median of the STRONG depths among self + 20 fixed
// opposite-colour offsets.
DVP_HD void filter_strong_px(const Dev& d, int px, int py) {
	const int W = d.width, H = d.height;
	const int center = py * W + px;
	if (d.costs[center] < 0.001f) return;
	float filter[21];
	int n = 0;
	filter[n++] = d.planes[center].w;
	// (dx, dy) in the reference's push order (APD.cu:3222-3284)
	const int ox[20] = { 0, 0, 0, 0, 0, 0, -1, -3, -5, 1, 3, 5, 2, 2, -2, -2, -1, 1, -1, 1 };
	const int oy[20] = { -1, -3, -5, 1, 3, 5, 0, 0, 0, 0, 0, 0, -1, 1, -1, 1, -2, -2, 2, 2 };
	for (int t = 0; t < 20; ++t) {
		const int x = px + ox[t], y = py + oy[t];
		if (x < 0 || y < 0 || x >= W || y >= H) continue;
		if (oy[t] == -2 && py <= 2) continue;   // the (+-1,-2) taps are guarded by p.y > 2 (APD.cu:3271,3275)
		const int q = x + y * W;
		if (d.weak_info[q] == DVP_STRONG) filter[n++] = d.planes[q].w;
	}
	sort_small(filter, n);
	const int m = n / 2;
	d.planes[center].w = (n % 2 == 0) ? (filter[m - 1] + filter[m]) / 2 : filter[m];
}

// shared prologue of DepthToWeak / LocalRefine (APD.cu:3928-3960, 4076-4108): cost at the current
// depth, mean baseline, weight sum over the selected views.
struct SweepCtx {
	f4 origin;          // camera-frame normal, .w = depth
	float depth;
	float cost_now, base_line, weight_normal;
	int valid;
	uint32_t sel;
};

// DepthToWeak (APD.cu:3892-4051).  REFINE: the same thread then does LocalRefine (APD.cu:4053-4139, the launch
// that follows, local_refine_px below) for its pixel.  LocalRefine's sweep slots -5..5 are planes the DepthToWeak
// sweep has just evaluated against the same views — same plane, same patch context, hence the same NCC and geometric
// costs bit for bit — so the fused kernel only folds them a second time in LocalRefine's own order
// ((ncc*w) + (factor*geom*w) per view) and adds LocalRefine's one extra slot, the current depth.  Neither kernel
// reads anything another pixel writes, so running them back to back per pixel is the two launches' result.
template <int SMP>
DVP_HD void local_refine_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals);
template <int SMP, bool REFINE>
DVP_HD void depth_to_weak_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	if (P.use_radius && d.radius[center] == 0) d.radius[center] = P.strong_radius;
	if (px < 6 || py < 6 || px >= W - 6 || py >= H - 6) {
		d.weak_info[center] = DVP_UNKNOWN;
		if (REFINE) local_refine_px<SMP>(d, px, py, tab, nevals);   // LocalRefine has no border rule
		return;
	}
	const f4 origin = normal_world_to_cam(rc, d.planes[center]);
	const float origin_depth = origin.w;
	if (origin_depth == 0) { d.weak_info[center] = DVP_UNKNOWN; return; }   // (LocalRefine returns here too)
	const uint32_t sel = d.selected_views[center];
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
	}
	float base_line = 0, weight_normal = 0.0f;
	int valid = 0;
	for (int v = 0; v < S; ++v) {
		if (!is_set(sel, v)) continue;
		weight_normal += vw[v];
		const float c0 = rc.c[0] - d.cameras[v + 1].c[0];
		const float c1 = rc.c[1] - d.cameras[v + 1].c[1];
		const float c2 = rc.c[2] - d.cameras[v + 1].c[2];
		base_line += sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
		valid++;
	}
	if (valid == 0) { d.weak_info[center] = DVP_UNKNOWN; return; }   // (LocalRefine returns here too)
	base_line /= valid;
	const float disp = rc.K[0] * base_line / origin_depth;
	const bool refine = REFINE && weight_normal != 0;   // LocalRefine's own guard (APD.cu:4075)
	// the reference also evaluates cost_now at the current depth (APD.cu:3937-3941); its value
	// is never used by DepthToWeak, so those evaluations are not issued.
	float p_costs[61];
	float lr_costs[11];            // LocalRefine's total of sweep slot pd = -5..5
	unsigned lr_live = 0;          // ... bit pd+5: the slot is inside the depth range
	// Sweep order: the central window first.  A pixel is WEAK whenever there is no cost peak <= 0.5 within
	// weak_peak_radius of the current depth — either there is no peak at all (min_peak = 0), or the lowest peak lies outside
	// the window, or it lies inside with a cost above 0.5 (APD.cu:4018-4023) — and that is decided by the window and its two
	// neighbours alone: such pixels skip the other ~46 planes (exact: their costs cannot change the state, and LocalRefine
	// only reads the slots -5..5).  Textureless regions, i.e. most WEAK pixels, leave here.  One loop, one call site of the
	// evaluator: k walks the window (-cw..cw), then the planes left of it, then those right of it.
	int cw = P.weak_peak_radius + 1;
	if (cw < 5) cw = 5;
	if (cw > 30) cw = 30;
	bool central_peak = false;
	auto window_has_peak = [&]() {
		bool any = false;
		for (int i = 30 - P.weak_peak_radius; i <= 30 + P.weak_peak_radius; ++i) {
			if (i < 2 || i > 58) continue;
			if (p_costs[i - 1] > p_costs[i] && p_costs[i + 1] > p_costs[i] && !(p_costs[i] > 0.5f)) any = true;
		}
		return any;
	};
	for (int k = 0; k < 61; ++k) {
		if (k == 2 * cw + 1) {
			central_peak = window_has_peak();
			if (!central_peak) break;
		}
		const int pd = k <= 2 * cw ? k - cw : (k - (2 * cw + 1) < 30 - cw ? k - (2 * cw + 1) - 30 : k - 30);
		const float p_depth = rc.K[0] * base_line / (disp + pd);
		if (p_depth < P.depth_min || p_depth > P.depth_max) { p_costs[pd + 30] = 2.0f; continue; }
		f4 pl = origin;
		pl.w = distance_to_origin(rc, px, py, p_depth, pl);
		const bool both = refine && pd >= -5 && pd <= 5;
		f3 fwd;   // the plane's 3-D point, shared by the views' geometric terms
		if (P.geom_consistency) fwd = geom_forward_point(d, px, py, pl);
		float pc = 0.0f, lr = 0.0f;
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			// a selected view with zero weight contributes cost*0 = +0: skipped (finite cost)
			if (vw[v] == 0) continue;
			const float ncc = ncc_old<SMP>(d, c, px, py, v + 1, pl);
			if (nevals) *nevals += 1;
			const float gc = P.geom_consistency ? geom_cost_of_point(d, px, py, v + 1, fwd) : 0.0f;
			float cst = ncc;
			if (P.geom_consistency) cst += P.geom_factor * gc;
			const float tc = 0.0f + cst;
			pc += tc * vw[v];
			if (both) {   // ncc*w and (factor*geom)*w added separately, APD.cu:4124-4126
				lr += ncc * vw[v];
				if (P.geom_consistency) lr += (P.geom_factor * gc * vw[v]);
			}
		}
		pc /= weight_normal;
		p_costs[pd + 30] = DVP_MIN(2.0f, pc);
		if (both) { lr_costs[pd + 5] = lr / weight_normal; lr_live |= 1u << (pd + 5); }
	}
	if (cw == 30) central_peak = window_has_peak();
	uint64_t is_peak = 0;
	int peak_count = 0, min_peak = 0;
	float min_cost = 2.0f;
	for (int i = 2; central_peak && i < 59; ++i) {
		if (p_costs[i - 1] > p_costs[i] && p_costs[i + 1] > p_costs[i]) {
			is_peak |= (uint64_t)1 << i;
			peak_count++;
			if (p_costs[i] < min_cost) { min_peak = i; min_cost = p_costs[i]; }
		}
	}
	const int dpk = min_peak - 30;
	uint8_t state;
	if (!central_peak) state = DVP_WEAK;
	else if ((dpk < 0 ? -dpk : dpk) > P.weak_peak_radius || p_costs[min_peak] > 0.5f) state = DVP_WEAK;
	else if (peak_count == 1) state = (p_costs[min_peak] <= 0.15f) ? DVP_STRONG : DVP_WEAK;
	else {
		float var = 0.0f;
		for (int i = 2; i < 59; ++i) {
			if (((is_peak >> i) & 1) && i != min_peak) {
				const float dist = p_costs[i] - min_cost;
				var += dist * dist;
			}
		}
		var = sqrtf(var);
		var /= (peak_count - 1);
		state = (var > 0.2f) ? DVP_STRONG : DVP_WEAK;
	}
	d.weak_info[center] = state;
	if (!refine) return;
	// ---- LocalRefine: the current depth (cost_now, APD.cu:4080-4090), then the minimum over the sweep slots ----
	float cost_now = 0.0f;
	{
		f4 pl = origin;
		pl.w = distance_to_origin(rc, px, py, origin_depth, pl);
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			if (vw[v] == 0) continue;
			const float ncc = ncc_old<SMP>(d, c, px, py, v + 1, pl);
			if (nevals) *nevals += 1;
			float t = ncc;   // (ncc + factor*geom) * w, APD.cu:4085-4089
			if (P.geom_consistency) t += P.geom_factor * geom_cost(d, px, py, v + 1, pl);
			cost_now += t * vw[v];
		}
		cost_now /= weight_normal;
	}
	float lr_min = 2.0f;
	int best_pd = -6;
	for (int pd = -5; pd <= 5; ++pd) {
		if (!((lr_live >> (pd + 5)) & 1)) continue;
		if (lr_costs[pd + 5] < lr_min) { lr_min = lr_costs[pd + 5]; best_pd = pd; }
	}
	const float best_depth = (best_pd == -6) ? origin_depth : rc.K[0] * base_line / (disp + best_pd);
	if (cost_now - lr_min > 0.1) d.planes[center].w = best_depth;
}

// LocalRefine (APD.cu:4053-4139)
template <int SMP>
DVP_HD void local_refine_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int W = d.width;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const f4 origin = normal_world_to_cam(rc, d.planes[center]);
	const float origin_depth = origin.w;
	if (origin_depth == 0) return;
	const uint32_t sel = d.selected_views[center];
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	float base_line = 0, weight_normal = 0.0f;
	int valid = 0;
	for (int v = 0; v < S; ++v) {
		if (!is_set(sel, v)) continue;
		weight_normal += vw[v];
		const float c0 = rc.c[0] - d.cameras[v + 1].c[0];
		const float c1 = rc.c[1] - d.cameras[v + 1].c[1];
		const float c2 = rc.c[2] - d.cameras[v + 1].c[2];
		base_line += sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
		valid++;
	}
	if (weight_normal == 0 || valid == 0) return;
	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
	}
	base_line /= valid;
	const float disp = rc.K[0] * base_line / origin_depth;
	float cost_now = 0.0f, min_cost = 2.0f, best_depth = origin_depth;
	// slot -6: the current depth (cost_now, APD.cu:4080-4090); slots -5..5: the sweep
	for (int pd = -6; pd <= 5; ++pd) {
		float p_depth = origin_depth;
		if (pd >= -5) {
			p_depth = rc.K[0] * base_line / (disp + pd);
			if (p_depth < P.depth_min || p_depth > P.depth_max) continue;
		}
		f4 pl = origin;
		pl.w = distance_to_origin(rc, px, py, p_depth, pl);
		float tc = 0.0f;
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			if (vw[v] == 0) continue;   // contributes +0 (finite costs)
			const float ncc = ncc_old<SMP>(d, c, px, py, v + 1, pl);
			if (nevals) *nevals += 1;
			const float gc = P.geom_consistency ? geom_cost(d, px, py, v + 1, pl) : 0.0f;
			if (pd == -6) {   // (ncc + factor*geom) * w, APD.cu:4085-4089
				float t = ncc;
				if (P.geom_consistency) t += P.geom_factor * gc;
				tc += t * vw[v];
			} else {          // ncc*w and (factor*geom)*w added separately, APD.cu:4124-4126
				tc += ncc * vw[v];
				if (P.geom_consistency) tc += (P.geom_factor * gc * vw[v]);
			}
		}
		tc /= weight_normal;
		if (pd == -6) cost_now = tc;
		else if (tc < min_cost) { min_cost = tc; best_depth = p_depth; }
	}
	if (cost_now - min_cost > 0.1) d.planes[center].w = best_depth;
}

// ---- DepthToWeak + LocalRefine as view-compacted evaluation passes --------------------------------------------------------
// depth_to_weak_px<SMP, true> above is the definition: one lane per pixel walks 61 (+1) planes x its selected views.  A wave
// of 64 pixels executes a view as soon as ONE lane selected it (6.4 views per pixel, 8.0 per wave at cfg3), every plane makes
// the workgroup cycle through all source images (125 B of HBM traffic per evaluation against 10.8 in dvp_strong_eval), and the
// cost line, the weights and the peak logic sit in the evaluator's register budget.  Same evaluations, same bits, as passes:
//   sweep_prepare_px   per pixel: validity, (camera-frame normal, depth), mean baseline, disparity, weight sum -> sweep_rec
//   sweep_eval_px      per (pixel, source view): the planes of one stage against ONE view — launched view by view over the
//                      pixels that selected the view with a weight > 0 (compacted tile by tile: dvp_sweep_eval), nothing live
//                      but the evaluator -> sweep_cost.  Stage 0: the central window (-cw..cw) and LocalRefine's extra slot
//                      (the current depth); stage 1: the rest of the line, only for pixels whose window holds a cost peak
//                      (the others are WEAK whatever the rest says: exact, see depth_to_weak_px)
//   sweep_decide1_px   per pixel: folds the window in the reference's view order, decides `central peak`, does LocalRefine
//                      (its slots -5..5 and the current depth are all stage 0)
//   sweep_decide2_px   per pixel with a central peak: folds the rest of the line, peak statistics -> weak_info
// Border pixels (DepthToWeak marks them UNKNOWN, LocalRefine has no border rule) go through the fused kernel.
constexpr int kSweepExtra = 72;     // sweep_cost field of the current depth: ncc + factor * geom (APD.cu:4085-4089)
constexpr int kSweepFields = 73;    // [0, 61): slot pd + 30 — ncc for |pd| <= 5, ncc + factor * geom otherwise; [61, 72): geom of slot |pd| <= 5
enum { SWF_VALID = 1, SWF_REFINE = 2, SWF_PEAK = 4 };
DVP_HD int sweep_window(const DvpParams& P) { int cw = P.weak_peak_radius + 1; if (cw < 5) cw = 5; if (cw > 30) cw = 30; return cw; }
// sweep_cost, the per (view, field, pixel) record.  DVP_SWEEP_LAYOUT 1 (default): [group of 64 pixels][view][field][64] — the 73 fields
// of a (group, view) are 18.7 KB in a row: an evaluation launch (one view) fills them slot after slot, a decision pass reads a view's 50
// slots from 50 neighbouring 256-byte lines.  0: [view][field][pixel] (rounds 4-5): the same slots lie L floats apart — 9 x 50 streams
// 100 MB from each other per wave (dvp_sweep_decide2 read its 48 GB at 2.4 TB/s, profiles/pmc_r06.json).
#ifndef DVP_SWEEP_LAYOUT
#define DVP_SWEEP_LAYOUT 1
#endif
DVP_HD size_t sweep_cost_floats(size_t L, int S) { return ((L + 63) / 64) * 64 * (size_t)S * kSweepFields; }
DVP_HD size_t sweep_field_stride(const Dev& d) {   // between field f and f + 1 of the same (view, pixel)
#if DVP_SWEEP_LAYOUT == 1
	(void)d;
	return 64;
#else
	return (size_t)d.width * (size_t)d.height;
#endif
}
DVP_HD size_t sweep_cost_index(const Dev& d, int v, int f, int center) {
#if DVP_SWEEP_LAYOUT == 1
	const int S = d.params.num_images - 1;
	const int rel = center - d.sweep_px0;   // (the band's first pixel; 0 when the buffer holds the whole image)
	return (((size_t)(rel >> 6) * S + v) * kSweepFields + (size_t)f) * 64 + (size_t)(rel & 63);
#else
	const size_t L = (size_t)d.width * (size_t)d.height;
	return ((size_t)v * kSweepFields + (size_t)f) * L + (size_t)center;
#endif
}
DVP_HD bool sweep_is_border(const Dev& d, int px, int py) { return px < 6 || py < 6 || px >= d.width - 6 || py >= d.height - 6; }

DVP_HD void sweep_prepare_px(const Dev& d, int px, int py) {
	const int W = d.width;
	const int center = px + py * W;
	const size_t L = (size_t)W * (size_t)d.height;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	if (P.use_radius && d.radius[center] == 0) d.radius[center] = P.strong_radius;
	f4 info = mk4(0.0f, 0.0f, 0.0f, 0.0f);   // .w: flags (as an integer value)
	d.sweep_rec[L + center] = info;
	if (sweep_is_border(d, px, py)) { d.weak_info[center] = DVP_UNKNOWN; return; }   // (+ LocalRefine: the fused kernel's border launch)
	const f4 origin = normal_world_to_cam(rc, d.planes[center]);
	if (origin.w == 0) { d.weak_info[center] = DVP_UNKNOWN; return; }
	const uint32_t sel = d.selected_views[center];
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	float base_line = 0, weight_normal = 0.0f;
	int valid = 0;
	for (int v = 0; v < S; ++v) {
		if (!is_set(sel, v)) continue;
		weight_normal += vw[v];
		const float c0 = rc.c[0] - d.cameras[v + 1].c[0];
		const float c1 = rc.c[1] - d.cameras[v + 1].c[1];
		const float c2 = rc.c[2] - d.cameras[v + 1].c[2];
		base_line += sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
		valid++;
	}
	if (valid == 0) { d.weak_info[center] = DVP_UNKNOWN; return; }
	base_line /= valid;
	const float disp = rc.K[0] * base_line / origin.w;
	d.sweep_rec[center] = origin;
	d.sweep_rec[L + center] = mk4(base_line, disp, weight_normal, (float)(SWF_VALID | (weight_normal != 0 ? SWF_REFINE : 0)));
}

// does (pixel, view v (0-based)) take part in stage `stage`?
DVP_HD bool sweep_go(const Dev& d, int center, int v, int stage) {
	const size_t L = (size_t)d.width * (size_t)d.height;
	const int flags = (int)d.sweep_rec[L + center].w;
	if (!(flags & (stage == 0 ? SWF_VALID : SWF_PEAK))) return false;
	return is_set(d.selected_views[center], v) && d.view_weight[(size_t)center * 32 + v] != 0;
}

// `cams`: the reference camera and the camera of view v + 1 ([0], [1]) where the caller keeps them (LDS), or null
template <int SMP>
DVP_HD void sweep_eval_px(const Dev& d, int px, int py, int v, int stage, PatchTab tab, unsigned long long* nevals, const DvpCamera* cams = nullptr) {
	const int W = d.width;
	const int center = px + py * W;
	const size_t L = (size_t)W * (size_t)d.height;
	const DvpParams& P = d.params;
	const DvpCamera rc_own = load_camera(d, 0);
	const DvpCamera& rc = cams ? cams[0] : rc_own;
	const f4 origin = d.sweep_rec[center];
	const f4 info = d.sweep_rec[L + center];
	const float base_line = info.x, disp = info.y;
	const bool refine = ((int)info.w & SWF_REFINE) != 0;
	PatchCtx c;
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
#if defined(DVP_ABL_SWEEP_NO_TABLE)   // timing ablations (wrong results): where the launch site's time goes, profiles/r06_sweep_ablation.txt
		c.tab = tab; c.radius = radius; c.inc = inc; c.fast = 1; c.sum_ref = 1.0f; c.sum_ref_ref = 2.0f; c.wsum = 1.0f;
		for (int t = 0; t < kTaps * kTaps; ++t) tab.set(t, mk2(d.nb_cos, d.nb_sin * (float)t));   // (written, so the evaluator stays alive)
#else
		build_patch_ctx(d, px, py, radius, inc, 0, tab, &c);
#endif
	}
	const int cw = sweep_window(P);
	float* out = d.sweep_cost + sweep_cost_index(d, v, 0, center);
	const size_t FS = sweep_field_stride(d);
	const int k0 = stage == 0 ? 0 : 2 * cw + 1, k1 = stage == 0 ? 2 * cw + 1 + (refine ? 1 : 0) : 61;
	for (int k = k0; k < k1; ++k) {
		const bool extra = stage == 0 && k == 2 * cw + 1;   // LocalRefine's current-depth slot
		const int pd = k <= 2 * cw ? k - cw : (k - (2 * cw + 1) < 30 - cw ? k - (2 * cw + 1) - 30 : k - 30);
		float p_depth = origin.w;
		if (!extra) {
			p_depth = rc.K[0] * base_line / (disp + pd);
			if (p_depth < P.depth_min || p_depth > P.depth_max) continue;
		}
		f4 pl = origin;
		pl.w = distance_to_origin(rc, px, py, p_depth, pl);
		// the geometric term's depth-map fetch is issued before the NCC evaluation and used after it
		GeomFetch gf;
		const bool split_geom = P.geom_consistency && cams != nullptr;
#if !defined(DVP_ABL_SWEEP_NO_GEOM) && !defined(DVP_SWEEP_GEOM_AFTER)
		if (split_geom) gf = geom_cost_fetch(d, cams[0], cams[1], v + 1, px, py, pl);
#endif
#if defined(DVP_ABL_SWEEP_NO_EVAL)
		const float ncc = pl.w * 1e-3f;
#else
		const float ncc = ncc_old<SMP>(d, c, px, py, v + 1, pl);
#endif
		if (nevals) *nevals += 1;
#if defined(DVP_ABL_SWEEP_NO_GEOM)   // timing ablation (wrong results)
		const float gc = 0.0f;
#elif defined(DVP_SWEEP_GEOM_AFTER)   // round 5's order (A/B)
		const float gc = !P.geom_consistency ? 0.0f : (cams ? geom_cost_cams(d, cams[0], cams[1], v + 1, px, py, pl) : geom_cost(d, px, py, v + 1, pl));
#else
		const float gc = !P.geom_consistency ? 0.0f : (split_geom ? geom_cost_finish(cams[0], cams[1], px, py, gf) : geom_cost(d, px, py, v + 1, pl));
#endif
#if defined(DVP_ABL_SWEEP_NO_STORE)
		if (ncc + gc != d.nb_thresh * 1e30f) continue;   // (never equal; a runtime value, so nothing is folded away)
#endif
		if (extra) {
			float t = ncc;
			if (P.geom_consistency) t += P.geom_factor * gc;
			DVP_NT_STORE(1, &out[(size_t)kSweepExtra * FS], t);
		} else if (pd >= -5 && pd <= 5) {
			DVP_NT_STORE(1, &out[(size_t)(pd + 30) * FS], ncc);
			if (P.geom_consistency) DVP_NT_STORE(1, &out[(size_t)(61 + pd + 5) * FS], gc);
		} else {
			float cst = ncc;
			if (P.geom_consistency) cst += P.geom_factor * gc;
			DVP_NT_STORE(1, &out[(size_t)(pd + 30) * FS], cst);
		}
	}
}

// The decision passes fold the per-view costs of a slot in the reference's view order (APD.cu:3965-4005, 4110-4130):
// pc = sum over the selected views of (0 + cost) * weight.  View outside, slots inside (unrolled): the ~50 loads of one
// view are in flight together, and every slot's sum still receives its terms in view order — a slot-outside loop is one
// dependent round trip per (slot, view) (35 ms per cfg3 launch for 33 GB).
// fold of sweep slot pd over the views in the reference's order (APD.cu:3965-4005, 4110-4130): *pc = the DepthToWeak cost
// (not yet divided), *lr = LocalRefine's total of the same slot
DVP_HD void sweep_fold(const Dev& d, int center, int pd, uint32_t sel, const uint8_t* vw, bool both, float* pc_out, float* lr_out) {
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	const size_t FS = sweep_field_stride(d);
	float pc = 0.0f, lr = 0.0f;
	for (int v = 0; v < S; ++v) {
		if (!is_set(sel, v)) continue;
		if (vw[v] == 0) continue;
		const float* in = d.sweep_cost + sweep_cost_index(d, v, 0, center);
		float ncc = 0.0f, gc = 0.0f, cst;
		if (pd >= -5 && pd <= 5) {
			ncc = DVP_NT_LOAD(2, &in[(size_t)(pd + 30) * FS]);
			cst = ncc;
			if (P.geom_consistency) { gc = DVP_NT_LOAD(2, &in[(size_t)(61 + pd + 5) * FS]); cst += P.geom_factor * gc; }
		} else {
			cst = DVP_NT_LOAD(2, &in[(size_t)(pd + 30) * FS]);
		}
		const float tc = 0.0f + cst;
		pc += tc * vw[v];
		if (both) {   // ncc*w and (factor*geom)*w added separately, APD.cu:4124-4126
			lr += ncc * vw[v];
			if (P.geom_consistency) lr += (P.geom_factor * gc * vw[v]);
		}
	}
	*pc_out = pc;
	*lr_out = lr;
}

DVP_HD void sweep_decide1_px(const Dev& d, int px, int py) {
	const int W = d.width;
	const int center = px + py * W;
	const size_t L = (size_t)W * (size_t)d.height;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const f4 info = d.sweep_rec[L + center];
	const int flags = (int)info.w;
	if (!(flags & SWF_VALID)) return;
	const f4 origin = d.sweep_rec[center];
	const float base_line = info.x, disp = info.y, weight_normal = info.z;
	const bool refine = (flags & SWF_REFINE) != 0;
	const uint32_t sel = d.selected_views[center];
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	const int cw = sweep_window(P);
	float lr_costs[11];
	unsigned lr_live = 0;
	float cost_now = 0.0f;
#if !defined(DVP_SWEEP_FOLD_BRANCHY)
	if (cw == 5) {
		// The default window: the 11 slots are LocalRefine's too.  View outside, the view's 23 fields fetched together, the sums as
		// selects — every slot's sum still receives its terms in view order (sweep_fold below, slot outside with a load per (slot, view)
		// inside two branches, is the same arithmetic: 150 dependent round trips per pixel, 8.3 ms for 25 GB at cfg3).
		const size_t FS = sweep_field_stride(d);
		float pc[11], lr[11];
		unsigned in_range = 0;
#pragma unroll
		for (int i = 0; i < 11; ++i) {
			pc[i] = 0.0f; lr[i] = 0.0f;
			const float p_depth = rc.K[0] * base_line / (disp + (i - 5));
			if (!(p_depth < P.depth_min || p_depth > P.depth_max)) in_range |= 1u << i;
		}
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			if (vw[v] == 0) continue;
			const float* in = d.sweep_cost + sweep_cost_index(d, v, 0, center);
			const float w = vw[v];
			float ncc[11], gc[11];
#pragma unroll
			for (int i = 0; i < 11; ++i) {
				ncc[i] = DVP_NT_LOAD(2, &in[(size_t)(25 + i) * FS]);
				gc[i] = 0.0f;
				if (P.geom_consistency) gc[i] = DVP_NT_LOAD(2, &in[(size_t)(61 + i) * FS]);
			}
			const float extra = DVP_NT_LOAD(2, &in[(size_t)kSweepExtra * FS]);
			sched_fence();
#pragma unroll
			for (int i = 0; i < 11; ++i) {
				float cst = ncc[i];
				if (P.geom_consistency) cst += P.geom_factor * gc[i];
				const float tc = 0.0f + cst;
				const float t = pc[i] + tc * w;
				pc[i] = ((in_range >> i) & 1) ? t : pc[i];
				float u = lr[i] + ncc[i] * w;   // ncc*w and (factor*geom)*w added separately, APD.cu:4124-4126
				if (P.geom_consistency) u += (P.geom_factor * gc[i] * w);
				lr[i] = ((in_range >> i) & 1) ? u : lr[i];
			}
			const float cn = cost_now + extra * w;
			cost_now = refine ? cn : cost_now;
		}
#pragma unroll
		for (int i = 0; i < 11; ++i) {
			float pcv = 2.0f;
			if ((in_range >> i) & 1) {
				pcv = DVP_MIN(2.0f, pc[i] / weight_normal);
				if (refine) { lr_costs[i] = lr[i] / weight_normal; lr_live |= 1u << i; }
			}
			d.sweep_pc[(size_t)(25 + i) * L + center] = pcv;
		}
	} else
#endif
	{
	for (int pd = -cw; pd <= cw; ++pd) {
		const float p_depth = rc.K[0] * base_line / (disp + pd);
		float pcv = 2.0f;
		if (!(p_depth < P.depth_min || p_depth > P.depth_max)) {
			const bool both = refine && pd >= -5 && pd <= 5;
			float pc, lr;
			sweep_fold(d, center, pd, sel, vw, both, &pc, &lr);
			pc /= weight_normal;
			pcv = DVP_MIN(2.0f, pc);
			if (both) { lr_costs[pd + 5] = lr / weight_normal; lr_live |= 1u << (pd + 5); }
		}
		d.sweep_pc[(size_t)(pd + 30) * L + center] = pcv;
	}
	if (refine)
		for (int v = 0; v < S; ++v) {
			if (!is_set(sel, v)) continue;
			if (vw[v] == 0) continue;
			cost_now += d.sweep_cost[sweep_cost_index(d, v, kSweepExtra, center)] * vw[v];
		}
	}
	bool central_peak = false;
	for (int i = 30 - P.weak_peak_radius; i <= 30 + P.weak_peak_radius; ++i) {
		if (i < 2 || i > 58) continue;
		const float a = d.sweep_pc[(size_t)(i - 1) * L + center], b = d.sweep_pc[(size_t)i * L + center], e = d.sweep_pc[(size_t)(i + 1) * L + center];
		if (a > b && e > b && !(b > 0.5f)) central_peak = true;
	}
	if (!central_peak) d.weak_info[center] = DVP_WEAK;
	else d.sweep_rec[L + center].w = (float)(flags | SWF_PEAK);
	if (!refine) return;
	// ---- LocalRefine: the current depth (cost_now, APD.cu:4080-4090: summed with the folds above), then the minimum over the sweep slots ----
	cost_now /= weight_normal;
	float lr_min = 2.0f;
	int best_pd = -6;
	for (int pd = -5; pd <= 5; ++pd) {
		if (!((lr_live >> (pd + 5)) & 1)) continue;
		if (lr_costs[pd + 5] < lr_min) { lr_min = lr_costs[pd + 5]; best_pd = pd; }
	}
	const float best_depth = (best_pd == -6) ? origin.w : rc.K[0] * base_line / (disp + best_pd);
	if (cost_now - lr_min > 0.1) d.planes[center].w = best_depth;
}

// (the window is 11 slots at the default radius: the slot-outside fold of sweep_fold measures 8 ms per cfg3 launch, the unrolled
// view-outside form used below for the rest of the line 13 — 168 registers and a predicate per slot)
// CW: the window half-width known at compile time (5 at the default weak_peak_radius), 0 = read from the parameters.  With a run-time
// window every slot's `inside the window?` is a scalar branch around its load, and a wait at every join.
template <int CW>
DVP_HD void sweep_decide2_cw(const Dev& d, int px, int py) {
	const int W = d.width;
	const int center = px + py * W;
	const size_t L = (size_t)W * (size_t)d.height;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const f4 info = d.sweep_rec[L + center];
	if (!((int)info.w & SWF_PEAK)) return;
	const float base_line = info.x, disp = info.y, weight_normal = info.z;
	const uint32_t sel = d.selected_views[center];
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	const int cw = CW > 0 ? CW : sweep_window(P);
	const size_t FS = sweep_field_stride(d);
	uint64_t live = 0;   // slots outside the window that are inside the depth range
	for (int pd = -30; pd <= 30; ++pd) {
		if (pd >= -cw && pd <= cw) continue;
		const float p_depth = rc.K[0] * base_line / (disp + pd);
		if (!(p_depth < P.depth_min || p_depth > P.depth_max)) live |= (uint64_t)1 << (pd + 30);
	}
	float p_costs[61];
#pragma unroll
	for (int i = 0; i < 61; ++i) p_costs[i] = 0.0f;
	for (int v = 0; v < S; ++v) {
		if (!is_set(sel, v)) continue;
		if (vw[v] == 0) continue;
		const float* in = d.sweep_cost + sweep_cost_index(d, v, 0, center);
		const float w = vw[v];
		// every slot of the view is fetched — a slot outside the depth range holds whatever an earlier launch left, its sum is not
		// used — and the sums are selects, not branches: with `if (live) { load; add }` the compiler kept each load inside its branch
		// with a wait behind it, 50 dependent round trips per view (20.0 ms for 48 GB at cfg3, r06_kernel_stats; DVP_SWEEP_FOLD_BRANCHY)
		float cst[61];
#pragma unroll
		for (int i = 0; i < 61; ++i) {
			const int pd = i - 30;
			cst[i] = 0.0f;
			if (pd >= -cw && pd <= cw) continue;          // (uniform; cw >= 5: the slots with separate ncc / geom fields are all inside)
#if defined(DVP_SWEEP_FOLD_BRANCHY)
			if ((live >> i) & 1)
#endif
			cst[i] = DVP_NT_LOAD(2, &in[(size_t)i * FS]);
		}
		sched_fence();
#pragma unroll
		for (int i = 0; i < 61; ++i) {
			const int pd = i - 30;
			if (pd >= -cw && pd <= cw) continue;
			const float tc = 0.0f + cst[i];
			const float t = p_costs[i] + tc * w;
			p_costs[i] = ((live >> i) & 1) ? t : p_costs[i];
		}
	}
#pragma unroll
	for (int i = 0; i < 61; ++i) {
		const int pd = i - 30;
		if (pd >= -cw && pd <= cw) { p_costs[i] = d.sweep_pc[(size_t)i * L + center]; continue; }
		float q = p_costs[i] / weight_normal;
		q = DVP_MIN(2.0f, q);
		p_costs[i] = ((live >> i) & 1) ? q : 2.0f;
	}
	uint64_t is_peak = 0;
	int peak_count = 0, min_peak = 0;
	float min_cost = 2.0f;
#pragma unroll
	for (int i = 2; i < 59; ++i) {
		if (p_costs[i - 1] > p_costs[i] && p_costs[i + 1] > p_costs[i]) {
			is_peak |= (uint64_t)1 << i;
			peak_count++;
			if (p_costs[i] < min_cost) { min_peak = i; min_cost = p_costs[i]; }
		}
	}
	// p_costs[min_peak] == min_cost whenever a peak below 2 exists; otherwise min_peak = 0 and the line's first entry is read
	const float at_min = (min_peak == 0) ? p_costs[0] : min_cost;
	const int dpk = min_peak - 30;
	uint8_t state;
	if ((dpk < 0 ? -dpk : dpk) > P.weak_peak_radius || at_min > 0.5f) state = DVP_WEAK;
	else if (peak_count == 1) state = (at_min <= 0.15f) ? DVP_STRONG : DVP_WEAK;
	else {
		float var = 0.0f;
#pragma unroll
		for (int i = 2; i < 59; ++i) {
			if (((is_peak >> i) & 1) && i != min_peak) {
				const float dist = p_costs[i] - min_cost;
				var += dist * dist;
			}
		}
		var = sqrtf(var);
		var /= (peak_count - 1);
		state = (var > 0.2f) ? DVP_STRONG : DVP_WEAK;
	}
	d.weak_info[center] = state;
}
DVP_HD void sweep_decide2_px(const Dev& d, int px, int py) {
	if (sweep_window(d.params) == 5) sweep_decide2_cw<5>(d, px, py);
	else sweep_decide2_cw<0>(d, px, py);
}

}  // namespace dvp
#endif
