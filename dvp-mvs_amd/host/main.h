// main.h — host-side vocabulary of the reference's public API (/root/reference/main.h:39-124)
// expressed on top of the C ABI (include/dvp_mvs.h) instead of OpenCV / Boost / CUDA headers:
//   Camera            is the ABI's DvpCamera itself (same field names, 112 bytes);
//   PatchMatchParams  is the ABI's DvpParams plus the reference's default values, so an
//                     APD can hand `&params` straight to dvp_set_params();
//   path              is std::filesystem::path (the reference uses boost::filesystem);
//   cv::Mat           is replaced by the small Mat of Mat.h.
// Names are the reference's so that code written against it compiles unchanged.
#ifndef DVP_HOST_MAIN_H_
#define DVP_HOST_MAIN_H_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dvp_mvs.h"
#include "Mat.h"

// Capacity constants (main.h:39-45).  The first four are fixed by the engine's buffer layouts.
constexpr int MAX_IMAGES = DVP_MAX_IMAGES;
constexpr int NEIGHBOUR_NUM = DVP_NEIGHBOUR_NUM;
constexpr int EDGE_NEIGH_NUM = DVP_EDGE_NEIGH_NUM;
constexpr int LAB_BOUNDARY_NUM = DVP_LAB_BOUNDARY_NUM;
constexpr int NUM_IMAGES = 4;                // default top_k
constexpr int MAX_SEARCH_RADIUS = 4096;

using path = std::filesystem::path;

// CUDA's vector PODs, as the reference's host code uses them.
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };

using Camera = DvpCamera;                    // main.h:58-67: K R t c height width depth_min depth_max

// main.h:74-84; numeric values are the ABI's.
enum RunState { FIRST_INIT = DVP_FIRST_INIT, REFINE_INIT = DVP_REFINE_INIT, REFINE_ITER = DVP_REFINE_ITER };
enum PixelState { WEAK = DVP_WEAK, STRONG = DVP_STRONG, UNKNOWN = DVP_UNKNOWN };

// main.h:86-112.  The flag members are uint8_t in the ABI (C has no bool); they convert to and
// from bool implicitly, `state` to and from RunState.
struct PatchMatchParams : DvpParams {
	PatchMatchParams() : DvpParams{} {
		max_iterations = 3;       num_images = 5;           top_k = NUM_IMAGES;
		sigma_spatial = 5.0f;     sigma_color = 3.0f;
		depth_min = 0.0f;         depth_max = 1.0f;
		strong_radius = 5;        strong_increment = 2;
		weak_radius = 5;          weak_increment = 5;
		weak_peak_radius = 2;     rotate_time = 4;
		ransac_threshold = 0.005f; geom_factor = 0.2f;
		geom_consistency = false; use_detail = false;
		use_APD = use_edge = use_limit = use_label = use_radius = true;
		state = FIRST_INIT;
	}
};
static_assert(sizeof(PatchMatchParams) == sizeof(DvpParams) && sizeof(DvpParams) == 76, "params layout");
static_assert(sizeof(Camera) == 112, "camera layout");

struct PointList { float3 coord, color; };   // main.h:69-72, one fused point of the .ply

// A Delaunay triangle of the sparse-point prior with the depth ratio at each corner (main.h:126-130).
struct Triangle {
	Point pt1, pt2, pt3;
	float rate1 = 0, rate2 = 0, rate3 = 0;
	Triangle(const Point a, const Point b, const Point c) : pt1(a), pt2(b), pt3(c) {}
};

// One reference view's work item (main.h:114-124).
struct Problem {
	int index = 0, ref_image_id = 0, iteration = 0;
	std::vector<int> src_image_ids;
	path dense_folder, result_folder;
	int scale_size = 1;
	bool show_medium_result = true;
	PatchMatchParams params;
};

#endif
