// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// C entry points (ctypes) + the launch emulation / kernel sequence of APD::RunPatchMatch
// (APD.cu:4406-4532).  Kernels are pure functions of the pre-launch state (the strong update
// reads neighbours from a snapshot), so rows are processed in parallel with OpenMP without
// changing results.
#include "ora_core.h"
#include "ora_kernels.h"
#include <chrono>
#include <cstdio>

using namespace ora;

namespace {

template <class F>
void launch_full(Ctx& h, F f) {   // grid_size_full / block 16x16 (APD.cu:4412-4419): every pixel once
	const int xb = (h.width + 31) / 32;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
	for (int y = 0; y < h.height; ++y)
		for (int b = 0; b < xb; ++b)
			for (int x = b * 32; x < h.width && x < b * 32 + 32; ++x) f(make_int2(x, y));
}

// grid_size_half / block 32x16 (APD.cu:4421-4428) with the pixel map of APD.cu:3093-3100:
// p.x = bx*32+tx, p.y = 2*(by*16+ty) + ((tx&1) ^ colour); the grid covers rows
// [0, 2*16*ceil((H/2)/16)), so for odd H with (H/2)%16==0 the last row is never visited.
template <class F>
void launch_half(Ctx& h, int colour /*0 = black, 1 = red*/, F f) {
	const int rows_half = ((h.height / 2) + 15) / 16 * 16;
	const int xb = (h.width + 31) / 32;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
	for (int yh = 0; yh < rows_half; ++yh)
		for (int b = 0; b < xb; ++b)
			for (int x = b * 32; x < h.width && x < b * 32 + 32; ++x) {
				int y = 2 * yh + ((x & 1) ^ colour);
				if (y >= h.height) continue;
				f(make_int2(x, y));
			}
}

void alloc_state(Ctx& h) {
	const size_t L = (size_t)h.width * h.height;
	const int S = h.num_images - 1;
	h.planes.assign(L, float4{0, 0, 0, 0});
	h.fit_planes.assign(L, float4{0, 0, 0, 0});
	h.costs.assign(L, 0.0f);
	h.selected_views.assign(L + h.width, 0u);
	h.view_weight.assign(L * MAX_IMAGES, 0);
	h.weak_info.assign(L, (uint8_t)STRONG);
	h.weak_reliable.assign(L, 0);
	h.weak_nearest_strong.assign(L, short2{-1, -1});
	h.neighbours_map.assign(L, 0);
	h.candidate.assign(L * (size_t)(S > 0 ? S : 1) * LAB_BOUNDARY_NUM, short2{0, 0});
	h.edge.assign(L, 0);
	h.edge_neigh.assign(L * EDGE_NEIGH_NUM, short2{-1, -1});
	h.label.assign(L, 0);
	h.radius.assign(L, 5);
	h.weak_count = 0;
}

}  // namespace

enum OraStage {
	ST_GEN_EDGE_INFORM = 0, ST_FIND_NEAREST_STRONG = 1, ST_GEN_NEIGHBOURS = 2, ST_NEIGHBOUR_UPDATE = 3,
	ST_RANDOM_INIT = 4, ST_STRONG_UPDATE = 5, ST_RANSAC_FIT = 6, ST_WEAK_UPDATE = 7,
	ST_GET_DEPTH_NORMAL = 8, ST_FILTER_STRONG = 9, ST_DEPTH_TO_WEAK = 10, ST_LOCAL_REFINE = 11,
};
enum OraBuffer {
	BUF_PLANES = 0, BUF_COSTS = 1, BUF_SELECTED_VIEWS = 2, BUF_VIEW_WEIGHT = 3, BUF_WEAK_INFO = 4,
	BUF_WEAK_RELIABLE = 5, BUF_WEAK_NEAREST_STRONG = 6, BUF_NEIGHBOURS_MAP = 7, BUF_NEIGHBOURS = 8,
	BUF_FIT_PLANES = 9, BUF_CANDIDATE = 10, BUF_EDGE = 11, BUF_EDGE_NEIGH = 12, BUF_LABEL = 13,
	BUF_LABEL_BOUNDARY = 14, BUF_COMPLEX = 15, BUF_RADIUS = 16,
};

extern "C" {

void* ora_create(int width, int height, int num_images) {
	Ctx* h = new Ctx();
	h->width = width; h->height = height; h->num_images = num_images;
	std::memset(h->cameras, 0, sizeof(h->cameras));
	std::memset(&h->params, 0, sizeof(h->params));
	alloc_state(*h);
	return h;
}
void ora_destroy(void* c) { delete (Ctx*)c; }

void ora_set_image(void* c, int idx, const float* data) { Ctx& h = *(Ctx*)c; h.images[idx].assign(data, data + (size_t)h.width * h.height); }
void ora_set_depth(void* c, int idx, const float* data) { Ctx& h = *(Ctx*)c; h.depths[idx].assign(data, data + (size_t)h.width * h.height); }
void ora_set_cameras(void* c, const Camera* cams, int n) { Ctx& h = *(Ctx*)c; for (int i = 0; i < n; ++i) h.cameras[i] = cams[i]; }
void ora_set_params(void* c, const PatchMatchParams* p) { Ctx& h = *(Ctx*)c; h.params = *p; }
void ora_set_seed(void* c, uint64_t seed) { ((Ctx*)c)->seed = seed; }
void ora_set_sampler(void* c, int sampler) { ((Ctx*)c)->sampler = sampler; }
// 0 = numerics contract (default), 1 = literal per-operator evaluation of the NCC expressions (KAT cross-check)
void ora_set_numerics(void* c, int numerics) { ((Ctx*)c)->numerics = numerics; literal_mode() = numerics; }
void ora_count_evals(void* c, int on) { Ctx& h = *(Ctx*)c; h.count_evals = on != 0; h.ncc_evals = 0; }
long long ora_get_evals(void* c) { return ((Ctx*)c)->ncc_evals; }

// input state (APD::CudaSpaceInitialization, APD.cpp:1560-1603): any pointer may be null.
// weak_info also rebuilds neighbours_map / weak_count (APD.cpp:1182-1193).
void ora_upload_state(void* c, const float4* planes, const uint32_t* views, const uint8_t* weak,
	const uint8_t* edge, const int* label, const int* radius) {
	Ctx& h = *(Ctx*)c;
	const size_t L = (size_t)h.width * h.height;
	if (planes) std::memcpy(h.planes.data(), planes, L * sizeof(float4));
	if (views) std::memcpy(h.selected_views.data(), views, L * sizeof(uint32_t));
	if (edge) std::memcpy(h.edge.data(), edge, L);
	if (label) std::memcpy(h.label.data(), label, L * sizeof(int));
	if (radius) std::memcpy(h.radius.data(), radius, L * sizeof(int));
	if (weak) std::memcpy(h.weak_info.data(), weak, L);
	h.weak_count = 0;
	for (size_t i = 0; i < L; ++i) {
		h.neighbours_map[i] = 0;
		if (h.weak_info[i] == WEAK) h.neighbours_map[i] = h.weak_count++;
	}
	const size_t wc = (size_t)(h.weak_count > 0 ? h.weak_count : 1);
	h.neighbours.assign(wc * NEIGHBOUR_NUM, short2{-1, -1});
	h.complex_.assign(wc, 0.0f);
	h.label_boundary.assign(wc * LAB_BOUNDARY_NUM, short2{-1, -1});
}

static void* buf_ptr(Ctx& h, int id, size_t* bytes) {
	switch (id) {
	case BUF_PLANES: *bytes = h.planes.size() * sizeof(float4); return h.planes.data();
	case BUF_COSTS: *bytes = h.costs.size() * 4; return h.costs.data();
	case BUF_SELECTED_VIEWS: *bytes = (size_t)h.width * h.height * 4; return h.selected_views.data();
	case BUF_VIEW_WEIGHT: *bytes = h.view_weight.size(); return h.view_weight.data();
	case BUF_WEAK_INFO: *bytes = h.weak_info.size(); return h.weak_info.data();
	case BUF_WEAK_RELIABLE: *bytes = h.weak_reliable.size(); return h.weak_reliable.data();
	case BUF_WEAK_NEAREST_STRONG: *bytes = h.weak_nearest_strong.size() * 4; return h.weak_nearest_strong.data();
	case BUF_NEIGHBOURS_MAP: *bytes = h.neighbours_map.size() * 4; return h.neighbours_map.data();
	case BUF_NEIGHBOURS: *bytes = h.neighbours.size() * 4; return h.neighbours.data();
	case BUF_FIT_PLANES: *bytes = h.fit_planes.size() * sizeof(float4); return h.fit_planes.data();
	case BUF_CANDIDATE: *bytes = h.candidate.size() * 4; return h.candidate.data();
	case BUF_EDGE: *bytes = h.edge.size(); return h.edge.data();
	case BUF_EDGE_NEIGH: *bytes = h.edge_neigh.size() * 4; return h.edge_neigh.data();
	case BUF_LABEL: *bytes = h.label.size() * 4; return h.label.data();
	case BUF_LABEL_BOUNDARY: *bytes = h.label_boundary.size() * 4; return h.label_boundary.data();
	case BUF_COMPLEX: *bytes = h.complex_.size() * 4; return h.complex_.data();
	case BUF_RADIUS: *bytes = h.radius.size() * 4; return h.radius.data();
	}
	*bytes = 0;
	return nullptr;
}
long long ora_buffer_bytes(void* c, int id) { size_t b; buf_ptr(*(Ctx*)c, id, &b); return (long long)b; }
int ora_get_buffer(void* c, int id, void* dst) {
	size_t b; void* p = buf_ptr(*(Ctx*)c, id, &b);
	if (!p) return -1;
	std::memcpy(dst, p, b);
	return 0;
}
int ora_set_buffer(void* c, int id, const void* src) {
	size_t b; void* p = buf_ptr(*(Ctx*)c, id, &b);
	if (!p) return -1;
	std::memcpy(p, src, b);
	return 0;
}
int ora_weak_count(void* c) { return ((Ctx*)c)->weak_count; }

// one kernel launch of APD.cu:4430-4505.  colour: 0 = Black*, 1 = Red* (half launches only).
int ora_run_stage(void* c, int stage, int iter, int colour) {
	Ctx& h = *(Ctx*)c;
	const int W = h.width;
	switch (stage) {
	case ST_GEN_EDGE_INFORM: launch_full(h, [&](int2 p) { GenEdgeInform_px(h, p); }); break;
	case ST_FIND_NEAREST_STRONG: launch_full(h, [&](int2 p) { FindNearestStrongPoint_px(h, p); }); break;
	case ST_GEN_NEIGHBOURS: launch_full(h, [&](int2 p) { GenNeighbours_px(h, p); }); break;
	case ST_NEIGHBOUR_UPDATE: launch_full(h, [&](int2 p) { NeigbourUpdate_px(h, p); }); break;
	case ST_RANDOM_INIT: launch_full(h, [&](int2 p) { RandomInitialization_px(h, p); }); break;
	case ST_STRONG_UPDATE:
		h.planes_snap = h.planes;
		h.costs_snap = h.costs;
		launch_half(h, colour, [&](int2 p) {   // APD.cu:3127-3165
			if (h.weak_info[p.x + p.y * W] == WEAK) return;
			CheckerboardPropagationStrong_px(h, p, iter);
		});
		break;
	case ST_RANSAC_FIT: launch_full(h, [&](int2 p) { RANSACToGetFitPlane_px(h, p, iter); }); break;
	case ST_WEAK_UPDATE:
		launch_half(h, colour, [&](int2 p) {   // APD.cu:3091-3125
			if (h.weak_info[p.x + p.y * W] == WEAK) CheckerboardPropagationWeak_px(h, p, iter);
		});
		break;
	case ST_GET_DEPTH_NORMAL: launch_full(h, [&](int2 p) { GetDepthandNormal_px(h, p); }); break;
	case ST_FILTER_STRONG:
		launch_half(h, colour, [&](int2 p) {   // APD.cu:3296-3328
			if (h.weak_info[p.x + p.y * W] != WEAK) CheckerboardFilterStrong_px(h, p);
		});
		break;
	case ST_DEPTH_TO_WEAK: launch_full(h, [&](int2 p) { DepthToWeak_px(h, p); }); break;
	case ST_LOCAL_REFINE: launch_full(h, [&](int2 p) { LocalRefine_px(h, p); }); break;
	default: return -1;
	}
	return 0;
}

// The per-pixel body of one launch on a LIST of pixels (px = x0, y0, x1, y1, ...): full-size parity on samples.  Every launch
// site is a per-pixel function of the pre-launch state (the strong update reads its neighbours from the snapshot taken here),
// so the listed pixels get exactly what the whole launch would give them.  Half launches keep the reference's pixel map
// (APD.cu:3093-3100, 4421-4428): a listed pixel the launch of this colour does not visit is left alone.  Returns the number of
// listed pixels the launch visits, -1 for an unknown stage.
int ora_run_stage_pixels(void* c, int stage, int iter, int colour, const int* px, int n) {
	Ctx& h = *(Ctx*)c;
	const int W = h.width, H = h.height;
	const int rows_half = ((H / 2) + 15) / 16 * 16;
	const bool half = stage == ST_STRONG_UPDATE || stage == ST_WEAK_UPDATE || stage == ST_FILTER_STRONG;
	if (stage < ST_GEN_EDGE_INFORM || stage > ST_LOCAL_REFINE) return -1;
	if (stage == ST_STRONG_UPDATE) { h.planes_snap = h.planes; h.costs_snap = h.costs; }
	int visited = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : visited)
	for (int i = 0; i < n; ++i) {
		const int2 p = make_int2(px[2 * i], px[2 * i + 1]);
		if (p.x < 0 || p.y < 0 || p.x >= W || p.y >= H) continue;
		if (half) {
			const int r = p.y - ((p.x & 1) ^ colour);
			if (r < 0 || (r & 1) || r / 2 >= rows_half) continue;
		}
		const bool weak = h.weak_info[p.x + p.y * W] == WEAK;
		++visited;
		switch (stage) {
		case ST_GEN_EDGE_INFORM: GenEdgeInform_px(h, p); break;
		case ST_FIND_NEAREST_STRONG: FindNearestStrongPoint_px(h, p); break;
		case ST_GEN_NEIGHBOURS: GenNeighbours_px(h, p); break;
		case ST_NEIGHBOUR_UPDATE: NeigbourUpdate_px(h, p); break;
		case ST_RANDOM_INIT: RandomInitialization_px(h, p); break;
		case ST_STRONG_UPDATE: if (!weak) CheckerboardPropagationStrong_px(h, p, iter); break;
		case ST_RANSAC_FIT: RANSACToGetFitPlane_px(h, p, iter); break;
		case ST_WEAK_UPDATE: if (weak) CheckerboardPropagationWeak_px(h, p, iter); break;
		case ST_GET_DEPTH_NORMAL: GetDepthandNormal_px(h, p); break;
		case ST_FILTER_STRONG: if (!weak) CheckerboardFilterStrong_px(h, p); break;
		case ST_DEPTH_TO_WEAK: DepthToWeak_px(h, p); break;
		case ST_LOCAL_REFINE: LocalRefine_px(h, p); break;
		}
	}
	return visited;
}

// APD::RunPatchMatch (APD.cu:4406-4532).  Returns seconds spent in the iteration loop
// (APD.cu:4478-4492) through *iter_seconds when non-null.
int ora_run_patchmatch(void* c, double* iter_seconds) {
	Ctx& h = *(Ctx*)c;
	ora_run_stage(c, ST_GEN_EDGE_INFORM, 0, 0);
	ora_run_stage(c, ST_FIND_NEAREST_STRONG, 0, 0);
	ora_run_stage(c, ST_GEN_NEIGHBOURS, 0, 0);
	ora_run_stage(c, ST_NEIGHBOUR_UPDATE, 0, 0);
	ora_run_stage(c, ST_RANDOM_INIT, 0, 0);
	auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < h.params.max_iterations; ++i) {
		ora_run_stage(c, ST_STRONG_UPDATE, i, 0);
		ora_run_stage(c, ST_STRONG_UPDATE, i, 1);
		ora_run_stage(c, ST_RANSAC_FIT, i, 0);
		ora_run_stage(c, ST_WEAK_UPDATE, i, 0);
		ora_run_stage(c, ST_WEAK_UPDATE, i, 1);
	}
	auto t1 = std::chrono::steady_clock::now();
	if (iter_seconds) *iter_seconds = std::chrono::duration<double>(t1 - t0).count();
	ora_run_stage(c, ST_GET_DEPTH_NORMAL, 0, 0);
	ora_run_stage(c, ST_FILTER_STRONG, 0, 0);
	ora_run_stage(c, ST_FILTER_STRONG, 0, 1);
	ora_run_stage(c, ST_DEPTH_TO_WEAK, 0, 0);
	ora_run_stage(c, ST_LOCAL_REFINE, 0, 0);
	return 0;
}

// function-level entry points for known-answer tests
float ora_expf(float x) { return dvp_expf_contract(x); }
uint32_t ora_rand_u32(uint64_t seed, uint32_t pixel, uint32_t site, uint32_t k) { return dvp_rand_u32(seed, pixel, site, k); }
float ora_tex_linear(const float* img, int W, int H, float x, float y, int sampler) { return tex_linear(img, W, H, x, y, sampler); }
void ora_homography(const Camera* ref, const Camera* src, const float* plane, float* H) {
	ComputeHomography(*ref, *src, float4{plane[0], plane[1], plane[2], plane[3]}, H);
}
float ora_ncc_old(void* c, int x, int y, int src_idx, const float* plane) {
	return ComputeBilateralNCCOld(make_int2(x, y), src_idx, float4{plane[0], plane[1], plane[2], plane[3]}, *(Ctx*)c);
}
float ora_ncc_new(void* c, int x, int y, int src_idx, const float* plane) {
	return ComputeBilateralNCCNew(make_int2(x, y), src_idx, float4{plane[0], plane[1], plane[2], plane[3]}, *(Ctx*)c);
}
float ora_geom_cost(void* c, int x, int y, int src_idx, const float* plane) {
	return ComputeGeomConsistencyCost(make_int2(x, y), src_idx, float4{plane[0], plane[1], plane[2], plane[3]}, *(Ctx*)c);
}
// batch: n (pixel, plane) pairs x all source views -> out[n * (num_images-1)]
void ora_eval_cost_vectors(void* c, const int* px, const float* planes, int n, float* out) {
	Ctx& h = *(Ctx*)c;
	const int S = h.num_images - 1;
#pragma omp parallel for schedule(dynamic, 64)
	for (int i = 0; i < n; ++i)
		for (int v = 0; v < S; ++v)
			out[(size_t)i * S + v] = ComputeBilateralNCCOld(make_int2(px[2 * i], px[2 * i + 1]), v + 1,
				float4{planes[4 * i], planes[4 * i + 1], planes[4 * i + 2], planes[4 * i + 3]}, h);
}

}  // extern "C"
