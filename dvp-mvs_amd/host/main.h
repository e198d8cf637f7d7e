// main.h — host-side types of the reference's public API (/root/reference/main.h:39-124),
// without OpenCV / Boost / CUDA: boost::filesystem::path -> std::filesystem::path, cv::Mat -> Mat
// (Mat.h).  Layouts of Camera and PatchMatchParams are byte-identical to the reference's PODs and
// to the C ABI's DvpCamera / DvpParams (include/dvp_mvs.h).
#ifndef _MAIN_H_
#define _MAIN_H_

#include <vector>
#include <string>
#include <iostream>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <iomanip>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <filesystem>

#include "../../include/dvp_mvs.h"
#include "Mat.h"

#define MAX_IMAGES 32
#define NEIGHBOUR_NUM 12
#define NUM_IMAGES 4
#define EDGE_NEIGH_NUM 8
#define LAB_BOUNDARY_NUM 8
#define MAX_SEARCH_RADIUS 4096

using std::filesystem::path;

struct float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
struct float2 { float x, y; };
struct int2 { int x, y; };

struct Camera {          // main.h:58-67
	float K[9];
	float R[9];
	float t[3];
	float c[3];
	int height;
	int width;
	float depth_min;
	float depth_max;
};
static_assert(sizeof(Camera) == sizeof(DvpCamera) && sizeof(Camera) == 112, "Camera layout");

struct PointList {       // main.h:69-72
	float3 coord;
	float3 color;
};

enum RunState { FIRST_INIT, REFINE_INIT, REFINE_ITER };   // main.h:74-78
enum PixelState { WEAK, STRONG, UNKNOWN };                // main.h:80-84

struct PatchMatchParams {   // main.h:86-112
	int max_iterations = 3;
	int num_images = 5;
	float sigma_spatial = 5.0f;
	float sigma_color = 3.0f;
	int top_k = 4;
	float depth_min = 0.0f;
	float depth_max = 1.0f;
	bool geom_consistency = false;
	int strong_radius = 5;
	int strong_increment = 2;
	int weak_radius = 5;
	int weak_increment = 5;
	bool use_APD = true;
	bool use_edge = true;
	bool use_limit = true;
	bool use_label = true;
	bool use_detail = false;
	bool use_radius = true;
	int weak_peak_radius = 2;
	int rotate_time = 4;
	float ransac_threshold = 0.005f;
	float geom_factor = 0.2f;
	RunState state;
};
static_assert(sizeof(PatchMatchParams) == sizeof(DvpParams) && sizeof(PatchMatchParams) == 76, "PatchMatchParams layout");

struct Problem {         // main.h:114-124
	int index;
	int ref_image_id;
	std::vector<int> src_image_ids;
	path dense_folder;
	path result_folder;
	int scale_size = 1;
	PatchMatchParams params;
	bool show_medium_result = true;
	int iteration;
};

#endif
