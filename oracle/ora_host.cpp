// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// Sequential restatements of the HOST stages next to the hot path for the parity tests of
// dvp-mvs_amd/host/: RunFusion (ETH variant, /root/reference/APD.cpp:1809-1960 with Get3DPointonWorld :500-523,
// ProjectCamera :536-546, GetAngle :1797-1806) and the ratio-map half of the Depth-Anything prior
// (APD.cpp:1210-1356 with calculateZ :30-49).  Plain arrays in, plain arrays out; no file I/O, no OpenCV.
//
// Toolchain-dependent arithmetic (stated, not pinned): APD.cpp calls sqrt / pow / fabs / exp unqualified.  Here they
// resolve to libstdc++'s overload set (float in -> float out; pow(float, int) -> double, as C++11 specifies), which is
// what a g++ build that sees <math.h> through the OpenCV / CUDA headers gets.
#include "ora_common.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
using ora::Camera;
using ora::float2;
using ora::float3;

float3 Get3DPointonWorld(const int x, const int y, const float depth, const Camera camera) {   // APD.cpp:500-523
	float3 pointX, tmpX;
	pointX.x = depth * (x - camera.K[2]) / camera.K[0];
	pointX.y = depth * (y - camera.K[5]) / camera.K[4];
	pointX.z = depth;
	tmpX.x = camera.R[0] * pointX.x + camera.R[3] * pointX.y + camera.R[6] * pointX.z;
	tmpX.y = camera.R[1] * pointX.x + camera.R[4] * pointX.y + camera.R[7] * pointX.z;
	tmpX.z = camera.R[2] * pointX.x + camera.R[5] * pointX.y + camera.R[8] * pointX.z;
	float3 C;   // recomputed from R and t in binary32 (NOT Camera::c, which ReadCamera accumulates in double)
	C.x = -(camera.R[0] * camera.t[0] + camera.R[3] * camera.t[1] + camera.R[6] * camera.t[2]);
	C.y = -(camera.R[1] * camera.t[0] + camera.R[4] * camera.t[1] + camera.R[7] * camera.t[2]);
	C.z = -(camera.R[2] * camera.t[0] + camera.R[5] * camera.t[1] + camera.R[8] * camera.t[2]);
	pointX.x = tmpX.x + C.x;
	pointX.y = tmpX.y + C.y;
	pointX.z = tmpX.z + C.z;
	return pointX;
}
void ProjectCamera(const float3 PointX, const Camera camera, float2& point, float& depth) {   // APD.cpp:536-546
	float3 tmp;
	tmp.x = camera.R[0] * PointX.x + camera.R[1] * PointX.y + camera.R[2] * PointX.z + camera.t[0];
	tmp.y = camera.R[3] * PointX.x + camera.R[4] * PointX.y + camera.R[5] * PointX.z + camera.t[1];
	tmp.z = camera.R[6] * PointX.x + camera.R[7] * PointX.y + camera.R[8] * PointX.z + camera.t[2];
	depth = camera.K[6] * tmp.x + camera.K[7] * tmp.y + camera.K[8] * tmp.z;
	point.x = (camera.K[0] * tmp.x + camera.K[1] * tmp.y + camera.K[2] * tmp.z) / depth;
	point.y = (camera.K[3] * tmp.x + camera.K[4] * tmp.y + camera.K[5] * tmp.z) / depth;
}
// acos as the fusion's numerics contract specifies it (the oracle's own statement of dvp-mvs_amd/csrc/dvp_fuse_math.hpp): binary32,
// one IEEE operation per step, three ranges — a rational R(z) ~ (asin(x) - x) / x^3 below 0.5, acos(x) = 2 asin(sqrt((1 - x) / 2))
// with a split square root above, mirrored below -0.5.  The reference calls libm's acos (APD.cpp:1800), whose low-order bits
// depend on the C library it is linked with; like it, this is < 1 ulp from the true value.
float acos_contract(float x) {
	const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
	const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
	uint32_t hx;
	std::memcpy(&hx, &x, 4);
	const uint32_t ix = hx & 0x7fffffffu;
	if (ix >= 0x3f800000u) {
		if (ix == 0x3f800000u) return (hx >> 31) ? pi + 2.0f * pio2_lo : 0.0f;
		return (x - x) / (x - x);
	}
	if (ix < 0x3f000000u) {
		if (ix <= 0x32800000u) return pio2_hi + pio2_lo;
		const float z = x * x;
		const float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
		return pio2_hi - (x - (pio2_lo - x * r));
	}
	if (hx >> 31) {
		const float z = (1.0f + x) * 0.5f;
		const float p = z * (pS0 + z * (pS1 + z * pS2)), q = 1.0f + z * qS1;
		const float s = sqrtf(z);
		const float w = (p / q) * s - pio2_lo;
		return pi - 2.0f * (s + w);
	}
	const float z = (1.0f - x) * 0.5f;
	const float s = sqrtf(z);
	uint32_t is;
	std::memcpy(&is, &s, 4);
	is &= 0xfffff000u;
	float df;
	std::memcpy(&df, &is, 4);
	const float c = (z - df * df) / (s + df);
	const float p = z * (pS0 + z * (pS1 + z * pS2)), q = 1.0f + z * qS1;
	const float w = (p / q) * s + c;
	return 2.0f * (df + w);
}
float GetAngle(const float* v1, const float* v2) {   // APD.cpp:1797-1806
	float dot_product = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
	float angle = acos_contract(dot_product);
	if (angle != angle) return 0.0f;
	return angle;
}
}  // namespace

extern "C" {

// RunFusion (APD.cpp:1809-1960) on in-memory maps.  Per view v (index = position in pair.txt order): camera (as the
// fusion sees it, i.e. already rescaled to the depth map), depth [rows*cols], normal [rows*cols*3], weak [rows*cols],
// bgr [rows*cols*3] (the colour image at the depth map's size), block [rows*cols] or null (blocks/mask_<id>.jpg).
// All views share rows x cols here.  src_index[v*max_src + j] = view index of the j-th source of v, -1 = end.
// Output: out_xyz [cap*3], out_bgr [cap*3] (floats, as PointList holds them); returns the number of points.
int ora_run_fusion(int num_images, int rows, int cols, const Camera* cameras, const float* const* depths, const float* const* normals,
                   const uint8_t* const* weaks, const uint8_t* const* images, const uint8_t* const* blocks, const int* src_index, int max_src,
                   float* out_xyz, float* out_bgr, int cap) {
	std::vector<std::vector<uint8_t>> masks(num_images, std::vector<uint8_t>((size_t)rows * cols, 0));
	int n_points = 0;
	for (int i = 0; i < num_images; ++i) {
		const int ref_index = i;
		int num_ngb = 0;
		while (num_ngb < max_src && src_index[i * max_src + num_ngb] >= 0) ++num_ngb;
		for (int r = 0; r < rows; ++r) {
			for (int c = 0; c < cols; ++c) {
				if (blocks && blocks[ref_index] && blocks[ref_index][(size_t)r * cols + c] < 128) continue;
				if (masks[ref_index][(size_t)r * cols + c] == 1) continue;
				float ref_depth = depths[ref_index][(size_t)r * cols + c];
				if (ref_depth <= 0.0) continue;
				const float* ref_normal = &normals[ref_index][((size_t)r * cols + c) * 3];
				float3 PointX = Get3DPointonWorld(c, r, ref_depth, cameras[ref_index]);
				float3 consistent_Point = PointX;
				int num_consistent = 0;
				float dynamic_consistency = 0.0f;
				std::vector<ora::int2> used_list(num_ngb, ora::make_int2(-1, -1));
				for (int j = 0; j < num_ngb; ++j) {
					int src = src_index[i * max_src + j];
					float2 point;
					float proj_depth;
					ProjectCamera(PointX, cameras[src], point, proj_depth);
					int src_r = int(point.y + 0.5f);
					int src_c = int(point.x + 0.5f);
					if (src_c >= 0 && src_c < cols && src_r >= 0 && src_r < rows) {
						if (masks[src][(size_t)src_r * cols + src_c] == 1) continue;
						float src_depth = depths[src][(size_t)src_r * cols + src_c];
						if (src_depth <= 0.0) continue;
						const float* src_normal = &normals[src][((size_t)src_r * cols + src_c) * 3];
						float3 tmp_X = Get3DPointonWorld(src_c, src_r, src_depth, cameras[src]);
						float2 tmp_pt;
						ProjectCamera(tmp_X, cameras[ref_index], tmp_pt, proj_depth);
						float reproj_error = (float)std::sqrt(std::pow(c - tmp_pt.x, 2) + std::pow(r - tmp_pt.y, 2));
						float relative_depth_diff = std::fabs(proj_depth - ref_depth) / ref_depth;
						float angle = GetAngle(ref_normal, src_normal);
						if (reproj_error < 2.0f && relative_depth_diff < 0.01f && angle < 0.174533f) {
							used_list[j].x = src_c;
							used_list[j].y = src_r;
							float tmp_index = reproj_error + 200 * relative_depth_diff + angle * 10;
							dynamic_consistency += ora::dvp_expf_contract(-tmp_index);   // exp by the contract (ora_common.h), as the engine's
							num_consistent++;
						}
					}
				}
				float factor = (weaks[ref_index][(size_t)r * cols + c] == ora::WEAK ? 0.45f : 0.3f);
				if (num_consistent >= 1 && (dynamic_consistency > factor * num_consistent)) {
					const uint8_t* px = &images[ref_index][((size_t)r * cols + c) * 3];
					float consistent_Color[3] = { (float)px[0], (float)px[1], (float)px[2] };
					for (int j = 0; j < num_ngb; ++j) {
						if (used_list[j].x == -1) continue;
						int src = src_index[i * max_src + j];
						masks[src][(size_t)used_list[j].y * cols + used_list[j].x] = 1;
						const uint8_t* color = &images[src][((size_t)used_list[j].y * cols + used_list[j].x) * 3];
						consistent_Color[0] += color[0];
						consistent_Color[1] += color[1];
						consistent_Color[2] += color[2];
					}
					consistent_Color[0] /= (num_consistent + 1);
					consistent_Color[1] /= (num_consistent + 1);
					consistent_Color[2] /= (num_consistent + 1);
					if (n_points < cap) {
						out_xyz[3 * n_points + 0] = consistent_Point.x;
						out_xyz[3 * n_points + 1] = consistent_Point.y;
						out_xyz[3 * n_points + 2] = consistent_Point.z;
						out_bgr[3 * n_points + 0] = consistent_Color[0];
						out_bgr[3 * n_points + 1] = consistent_Color[1];
						out_bgr[3 * n_points + 2] = consistent_Color[2];
					}
					n_points++;
				}
			}
		}
	}
	return n_points;
}

// RunFusion_TAT_Intermediate / RunFusion_TAT_advanced (APD.cpp:1962-2130 / 2132-2279) on in-memory maps; arguments as
// ora_run_fusion (no weak maps: these variants do not read them).  Differences to RunFusion that the restatement keeps: a
// reference pixel is NOT skipped for being claimed; the per-source residuals `diff[]` are one array per VIEW, only
// overwritten when a source yields a comparison (a source that drops out keeps the residuals of the last pixel it was
// compared for); a pixel is kept for the first k = 2..num_ngb with at least k sources inside k-scaled thresholds; a
// kept pixel claims ITSELF; the intermediate variant also tests the normals and averages the colours, the advanced one
// keeps the pixel's colour.
int ora_run_fusion_tat(int advanced, int num_images, int rows, int cols, const Camera* cameras, const float* const* depths, const float* const* normals,
                       const uint8_t* const* images, const uint8_t* const* blocks, const int* src_index, int max_src,
                       float* out_xyz, float* out_bgr, int cap) {
	const float dist_base = 0.25f;
	const float depth_base = advanced ? 1.0f / 3000.0f : 1.0f / 3500.0f;
	const float angle_base = 0.06981317007977318f;   // 4 degree
	const float angle_grad = 0.05235987755982988f;   // 3 degree
	struct CostData {
		float dist, depth, angle;
		int src_r, src_c;
		bool use;
		CostData() { dist = FLT_MAX; depth = FLT_MAX; angle = FLT_MAX; src_r = 0; src_c = 0; use = false; }
	};
	std::vector<std::vector<uint8_t>> masks(num_images, std::vector<uint8_t>((size_t)rows * cols, 0));
	int n_points = 0;
	for (int i = 0; i < num_images; ++i) {
		const int ref_index = i;
		int num_ngb = 0;
		while (num_ngb < max_src && src_index[i * max_src + num_ngb] >= 0) ++num_ngb;
		std::vector<CostData> diff(num_ngb, CostData());
		for (int r = 0; r < rows; ++r) {
			for (int c = 0; c < cols; ++c) {
				if (blocks && blocks[ref_index] && blocks[ref_index][(size_t)r * cols + c] < 128) continue;
				float ref_depth = depths[ref_index][(size_t)r * cols + c];
				if (ref_depth <= 0.0) continue;
				const float* ref_normal = &normals[ref_index][((size_t)r * cols + c) * 3];
				float3 PointX = Get3DPointonWorld(c, r, ref_depth, cameras[ref_index]);
				float3 consistent_Point = PointX;
				const uint8_t* px = &images[ref_index][((size_t)r * cols + c) * 3];
				for (int j = 0; j < num_ngb; ++j) {
					int src = src_index[i * max_src + j];
					float2 point;
					float proj_depth;
					ProjectCamera(PointX, cameras[src], point, proj_depth);
					int src_r = int(point.y + 0.5f);
					int src_c = int(point.x + 0.5f);
					if (src_c >= 0 && src_c < cols && src_r >= 0 && src_r < rows) {
						if (masks[src][(size_t)src_r * cols + src_c] == 1) continue;
						float src_depth = depths[src][(size_t)src_r * cols + src_c];
						if (src_depth <= 0.0) continue;
						const float* src_normal = &normals[src][((size_t)src_r * cols + src_c) * 3];
						float3 tmp_X = Get3DPointonWorld(src_c, src_r, src_depth, cameras[src]);
						float2 tmp_pt;
						ProjectCamera(tmp_X, cameras[ref_index], tmp_pt, proj_depth);
						float reproj_error = (float)std::sqrt(std::pow(c - tmp_pt.x, 2) + std::pow(r - tmp_pt.y, 2));
						float relative_depth_diff = std::fabs(proj_depth - ref_depth) / ref_depth;
						float angle = GetAngle(ref_normal, src_normal);
						diff[j].dist = reproj_error;
						diff[j].depth = relative_depth_diff;
						diff[j].angle = angle;
						diff[j].src_r = src_r;
						diff[j].src_c = src_c;
					}
				}
				for (int k = 2; k <= num_ngb; ++k) {
					int count = 0;
					for (int j = 0; j < num_ngb; ++j) {
						diff[j].use = false;
						if (diff[j].dist < k * dist_base && diff[j].depth < k * depth_base && (advanced || diff[j].angle < (k * angle_grad + angle_base))) {
							count++;
							diff[j].use = true;
						}
					}
					if (count >= k) {
						float consistent_Color[3] = { (float)px[0], (float)px[1], (float)px[2] };
						if (!advanced) {
							for (int j = 0; j < num_ngb; ++j) {
								if (diff[j].use) {
									int src = src_index[i * max_src + j];
									const uint8_t* color = &images[src][((size_t)diff[j].src_r * cols + diff[j].src_c) * 3];
									consistent_Color[0] += (float)color[0];
									consistent_Color[1] += (float)color[1];
									consistent_Color[2] += (float)color[2];
								}
							}
							consistent_Color[0] /= (count + 1.0f);
							consistent_Color[1] /= (count + 1.0f);
							consistent_Color[2] /= (count + 1.0f);
						}
						if (n_points < cap) {
							out_xyz[3 * n_points + 0] = consistent_Point.x;
							out_xyz[3 * n_points + 1] = consistent_Point.y;
							out_xyz[3 * n_points + 2] = consistent_Point.z;
							out_bgr[3 * n_points + 0] = consistent_Color[0];
							out_bgr[3 * n_points + 1] = consistent_Color[1];
							out_bgr[3 * n_points + 2] = consistent_Color[2];
						}
						n_points++;
						masks[ref_index][(size_t)r * cols + c] = 1;
						break;
					}
				}
			}
		}
	}
	return n_points;
}

}  // extern "C"
