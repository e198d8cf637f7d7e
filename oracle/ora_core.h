// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// Context, software sampler, small math helpers and the two NCC cost functions.
#ifndef ORA_CORE_H_
#define ORA_CORE_H_

#include "ora_common.h"
#include <vector>

namespace ora {

// The buffer bundle every kernel receives — restates DataPassHelper (APD.h:60-92) with the
// textures replaced by plain row-major float images and curandState by (seed).
struct Ctx {
	int width = 0, height = 0;
	int num_images = 0;
	int sampler = 0;   // 0 = cuda8 (8-bit fractional weights), 1 = exact
	int numerics = 0;  // 0 = the numerics contract (ora_common.h); 1 = literal per-operator evaluation of the reference's NCC expressions (KAT cross-check only)
	uint64_t seed = 0;
	PatchMatchParams params{};
	Camera cameras[MAX_IMAGES];
	std::vector<float> images[MAX_IMAGES];   // texture_objects_cuda
	std::vector<float> depths[MAX_IMAGES];   // texture_depths_cuda (geom_consistency only)
	std::vector<float4> planes;              // plane_hypotheses_cuda
	std::vector<float4> fit_planes;          // fit_plane_hypotheses_cuda (zeroed, APD.cpp:1571)
	std::vector<float> costs;                // costs_cuda
	std::vector<uint32_t> selected_views;    // selected_views_cuda (+ width zeroed tail: the
	                                         //  reference reads one row past the end, APD.cu:2473)
	std::vector<uint8_t> view_weight;        // view_weight_cuda, 32 per pixel
	std::vector<uint8_t> weak_info;          // weak_info_cuda
	std::vector<uint8_t> weak_reliable;      // weak_reliable_cuda
	std::vector<short2> weak_nearest_strong; // weak_nearest_strong
	std::vector<int> neighbours_map;         // neighbours_map_cuda (running index of WEAK pixels)
	std::vector<short2> neighbours;          // neighbours_cuda, 12 per WEAK pixel
	int weak_count = 0;
	std::vector<short2> candidate;           // candidate_cuda: [pixel][view][8]  (the reference's
	                                         //  stride is 4 views, main.h:41, and aliases for S>4)
	std::vector<uint8_t> edge;               // edge_cuda
	std::vector<short2> edge_neigh;          // edge_neigh_cuda, 8 per pixel
	std::vector<float> complex_;             // complex_cuda, per WEAK pixel
	std::vector<int> label;                  // label_cuda
	std::vector<short2> label_boundary;      // label_boundary_cuda, 8 per WEAK pixel
	std::vector<int> radius;                 // radius_cuda
	// snapshot of (planes, costs) taken before each strong red/black launch: defines the
	// same-colour reads of direction 4 (APD.cu:2071-2074 race) as "value before the launch".
	std::vector<float4> planes_snap;
	std::vector<float> costs_snap;
	long long ncc_evals = 0;                 // instrumentation (counted when count_evals)
	bool count_evals = false;
};

// ---- software texture (APD.cpp:1501-1517 semantics) ------------------------------------------
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tex2D(img, ix + 0.5f, iy + 0.5f) with integer ix,iy: exact texel, clamp-to-edge
inline float tex_texel(const float* img, int W, int H, int ix, int iy) {
	return img[clampi(iy, 0, H - 1) * W + clampi(ix, 0, W - 1)];
}

// The reference's tex2D(img, x + 0.5f, y + 0.5f) with cudaFilterModeLinear, unnormalised
// coordinates, clamp (APD.cpp:1501-1517), as a function of the PIXEL coordinate (x, y): the texture
// unit samples at (coordinate - 0.5), which cancels the +0.5f of the call sites (contract item in
// ora_common.h).  sampler 0 ("cuda8"): coordinate -> fixed point with 8 fractional bits, round
// half up (CUDA Programming Guide, "Linear Filtering": 9-bit fixed point with 8 bits of fraction);
// sampler 1: exact floor / fraction.
inline float tex_linear(const float* img, int W, int H, float x, float y, int sampler) {
	const float xb = fminf(fmaxf(x, -1.0f), (float)W);   // also maps NaN to -1
	const float yb = fminf(fmaxf(y, -1.0f), (float)H);
	int i0, j0;
	float a, b;
	if (sampler == 0) {
		const int qx = (int)floorf(fmaf(xb, 256.0f, 0.5f));
		const int qy = (int)floorf(fmaf(yb, 256.0f, 0.5f));
		i0 = qx >> 8;   // arithmetic shift: floor division by 256
		j0 = qy >> 8;
		a = (float)(qx & 255) * (1.0f / 256.0f);
		b = (float)(qy & 255) * (1.0f / 256.0f);
	} else {
		const float fx = floorf(xb), fy = floorf(yb);
		a = xb - fx;
		b = yb - fy;
		i0 = (int)fx;
		j0 = (int)fy;
	}
	const int x0 = clampi(i0, 0, W - 1), x1 = clampi(i0 + 1, 0, W - 1);
	const int y0 = clampi(j0, 0, H - 1), y1 = clampi(j0 + 1, 0, H - 1);
	const float t00 = img[y0 * W + x0], t10 = img[y0 * W + x1];
	const float t01 = img[y1 * W + x0], t11 = img[y1 * W + x1];
	const float top = fmaf(a, t10 - t00, t00);
	const float bot = fmaf(a, t11 - t01, t01);
	return fmaf(b, bot - top, top);
}

// Literal variant used by numerics == 1: the call sites' (x + 0.5f), the unit's (coordinate - 0.5f),
// both rounded in binary32, fractions rounded to 8 bits after the floor.
inline float tex_linear_literal(const float* img, int W, int H, float x, float y, int sampler) {
	float xb = x - 0.5f, yb = y - 0.5f;
	xb = fminf(fmaxf(xb, -1.0f), (float)W);
	yb = fminf(fmaxf(yb, -1.0f), (float)H);
	const float fx = floorf(xb), fy = floorf(yb);
	float a = xb - fx, b = yb - fy;
	if (sampler == 0) {
		a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
		b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
	}
	const int i0 = (int)fx, j0 = (int)fy;
	const int x0 = clampi(i0, 0, W - 1), x1 = clampi(i0 + 1, 0, W - 1);
	const int y0 = clampi(j0, 0, H - 1), y1 = clampi(j0 + 1, 0, H - 1);
	const float t00 = img[y0 * W + x0], t10 = img[y0 * W + x1];
	const float t01 = img[y1 * W + x0], t11 = img[y1 * W + x1];
	const float top = fmaf(a, t10 - t00, t00);
	const float bot = fmaf(a, t11 - t01, t01);
	return fmaf(b, bot - top, top);
}

// Reciprocals of up to 6 projective denominators with ONE correctly rounded division (contract:
// the projective divide of a patch row is taken six taps at a time): prefix products, 1/product,
// then peel the factors off again.  Every step is a plain binary32 multiply.
inline void batch_rcp(const float* z, int n, float* iz) {
	float p[6];
	p[0] = z[0];
	for (int k = 1; k < n; ++k) p[k] = p[k - 1] * z[k];
	float r = 1.0f / p[n - 1];
	for (int k = n - 1; k >= 1; --k) {
		iz[k] = r * p[k - 1];
		r = r * z[k];
	}
	iz[0] = r;
}

// ---- small helpers (APD.cu:3-499) -------------------------------------------------------------
inline void matMul3x3(const float* A, const float* B, float* C) {   // APD.cu:3-12
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			C[i * 3 + j] = 0;
			for (int k = 0; k < 3; ++k) C[i * 3 + j] += A[i * 3 + k] * B[k * 3 + j];
		}
}
inline void matMul3x1_ref(const float* A, const float* B, float* C) {  // APD.cu:14-18 (A[7] twice: kept)
	C[0] = A[0] * B[0] + A[1] * B[1] + A[2] * B[2];
	C[1] = A[3] * B[0] + A[4] * B[1] + A[5] * B[2];
	C[2] = A[6] * B[0] + A[7] * B[1] + A[7] * B[2];
}
inline void matTranspose3x3(const float* A, float* At) {            // APD.cu:20-26
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) At[j * 3 + i] = A[i * 3 + j];
}
inline void sort_small(float* d, const int n) {                     // APD.cu:114-123
	int j;
	for (int i = 1; i < n; i++) {
		float tmp = d[i];
		for (j = i; j >= 1 && tmp < d[j - 1]; j--) d[j] = d[j - 1];
		d[j] = tmp;
	}
}
inline void sort_small_weighted(short2* points, float* w, int n) {  // APD.cu:125-138
	int j;
	for (int i = 1; i < n; i++) {
		short2 tmp = points[i];
		float tmp_w = w[i];
		for (j = i; j >= 1 && tmp_w < w[j - 1]; j--) {
			points[j] = points[j - 1];
			w[j] = w[j - 1];
		}
		points[j] = tmp;
		w[j] = tmp_w;
	}
}
inline int FindMinCostIndex(const float* costs, const int n) {      // APD.cu:155-166 (ties -> last)
	float min_cost = costs[0];
	int min_cost_idx = 0;
	for (int idx = 1; idx < n; ++idx)
		if (costs[idx] <= min_cost) { min_cost = costs[idx]; min_cost_idx = idx; }
	return min_cost_idx;
}
inline void setBit(uint32_t* input, const unsigned n) { (*input) |= (uint32_t)(1u << n); }   // :181
inline void unSetBit(uint32_t* input, const unsigned n) { (*input) &= (uint32_t)(0xFFFFFFFEu << n); } // :186 (clears 0..n: kept)
inline int isSet(uint32_t input, const unsigned n) { return (input >> n) & 1; }               // :191

inline float Vec3DotVec3(const float3 a, const float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float Vec3DotVec3(const float3 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float Vec2CrossVec2(float2 a, float2 b) { return a.x * b.y - a.y * b.x; }   // APD.cu:232
inline float Vec2DotVec2(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }     // APD.cu:228

inline void NormalizeVec3(float4* v) {   // APD.cu:331-338 (rsqrtf -> 1/sqrtf)
	const float n2 = v->x * v->x + v->y * v->y + v->z * v->z;
	const float inv = 1.0f / sqrtf(n2);
	v->x *= inv; v->y *= inv; v->z *= inv;
}
inline void NormalizeVec2(float2* v) {   // APD.cu:348-354
	const float n2 = v->x * v->x + v->y * v->y;
	const float inv = 1.0f / sqrtf(n2);
	v->x *= inv; v->y *= inv;
}

inline void TransformPDFToCDF(float* probs, const int num_probs) {   // APD.cu:356-370
	float prob_sum = 0.0f;
	for (int i = 0; i < num_probs; ++i) prob_sum += probs[i];
	const float inv_prob_sum = 1.0f / prob_sum;
	float cum_prob = 0.0f;
	for (int i = 0; i < num_probs; ++i) {
		const float prob = probs[i] * inv_prob_sum;
		cum_prob += prob;
		probs[i] = cum_prob;
	}
}

template <class P>
inline void Get3DPoint(const Camera& camera, const P p, const float depth, float* X) {  // APD.cu:372-384
	X[0] = depth * (p.x - camera.K[2]) / camera.K[0];
	X[1] = depth * (p.y - camera.K[5]) / camera.K[4];
	X[2] = depth;
}
inline float4 GetViewDirection(const Camera& camera, const int2 p, const float depth) {   // APD.cu:386-398
	float X[3];
	Get3DPoint(camera, p, depth, X);
	float norm = sqrtf(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
	return make_float4(X[0] / norm, X[1] / norm, X[2] / norm, 0);
}
inline float GetDistance2Origin(const Camera& camera, const int2 p, const float depth, const float4 normal) {  // APD.cu:400-405
	float X[3];
	Get3DPoint(camera, p, depth, X);
	return -(normal.x * X[0] + normal.y * X[1] + normal.z * X[2]);
}
inline float ComputeDepthfromPlaneHypothesis(const Camera& camera, const float4 pl, const int2 p) {  // APD.cu:419-422
	return -pl.w * camera.K[0] / ((p.x - camera.K[2]) * pl.x + (camera.K[0] / camera.K[4]) * (p.y - camera.K[5]) * pl.y + camera.K[0] * pl.z);
}
inline float3 Get3DPointonWorld_cu(const float x, const float y, const float depth, const Camera& camera) {  // APD.cu:467-487
	float3 pointX, tmpX;
	pointX.x = depth * (x - camera.K[2]) / camera.K[0];
	pointX.y = depth * (y - camera.K[5]) / camera.K[4];
	pointX.z = depth;
	tmpX.x = camera.R[0] * pointX.x + camera.R[3] * pointX.y + camera.R[6] * pointX.z;
	tmpX.y = camera.R[1] * pointX.x + camera.R[4] * pointX.y + camera.R[7] * pointX.z;
	tmpX.z = camera.R[2] * pointX.x + camera.R[5] * pointX.y + camera.R[8] * pointX.z;
	pointX.x = tmpX.x + camera.c[0];
	pointX.y = tmpX.y + camera.c[1];
	pointX.z = tmpX.z + camera.c[2];
	return pointX;
}
inline void ProjectonCamera_cu(const float3 PointX, const Camera& camera, float2& point, float& depth) {  // APD.cu:489-499
	float3 tmp;
	tmp.x = camera.R[0] * PointX.x + camera.R[1] * PointX.y + camera.R[2] * PointX.z + camera.t[0];
	tmp.y = camera.R[3] * PointX.x + camera.R[4] * PointX.y + camera.R[5] * PointX.z + camera.t[1];
	tmp.z = camera.R[6] * PointX.x + camera.R[7] * PointX.y + camera.R[8] * PointX.z + camera.t[2];
	depth = camera.K[6] * tmp.x + camera.K[7] * tmp.y + camera.K[8] * tmp.z;
	point.x = (camera.K[0] * tmp.x + camera.K[1] * tmp.y + camera.K[2] * tmp.z) / depth;
	point.y = (camera.K[3] * tmp.x + camera.K[4] * tmp.y + camera.K[5] * tmp.z) / depth;
}
inline float4 TransformNormal(const Camera& camera, float4 pl) {          // APD.cu:750-758 (R^T n)
	float4 r;
	r.x = camera.R[0] * pl.x + camera.R[3] * pl.y + camera.R[6] * pl.z;
	r.y = camera.R[1] * pl.x + camera.R[4] * pl.y + camera.R[7] * pl.z;
	r.z = camera.R[2] * pl.x + camera.R[5] * pl.y + camera.R[8] * pl.z;
	r.w = pl.w;
	return r;
}
inline float4 TransformNormal2RefCam(const Camera& camera, float4 pl) {   // APD.cu:760-768 (R n)
	float4 r;
	r.x = camera.R[0] * pl.x + camera.R[1] * pl.y + camera.R[2] * pl.z;
	r.y = camera.R[3] * pl.x + camera.R[4] * pl.y + camera.R[5] * pl.z;
	r.z = camera.R[6] * pl.x + camera.R[7] * pl.y + camera.R[8] * pl.z;
	r.w = pl.w;
	return r;
}

// APD.cu:244-265
inline bool PointinTriangle(short2 A, short2 B, short2 C, int2 P) {
	float2 AB = make_float2(B.x - A.x, B.y - A.y);
	float2 BC = make_float2(C.x - B.x, C.y - B.y);
	float2 CA = make_float2(A.x - C.x, A.y - C.y);
	float AB_ = sqrtf(AB.x * AB.x + AB.y * AB.y);
	float BC_ = sqrtf(BC.x * BC.x + BC.y * BC.y);
	float CA_ = sqrtf(CA.x * CA.x + CA.y * CA.y);
	if (AB_ <= 2 || BC_ <= 2 || CA_ <= 2) return false;
	if (!(AB_ + BC_ > CA_ && BC_ + CA_ > AB_ && AB_ + CA_ > BC_)) return false;
	float2 PA = make_float2(A.x - P.x, A.y - P.y);
	float2 PB = make_float2(B.x - P.x, B.y - P.y);
	float2 PC = make_float2(C.x - P.x, C.y - P.y);
	float t1 = Vec2CrossVec2(PA, PB);
	float t2 = Vec2CrossVec2(PB, PC);
	float t3 = Vec2CrossVec2(PC, PA);
	return t1 * t2 >= 0 && t1 * t3 >= 0;
}

// APD.cu:267-311.  true = the segment B->A crosses an edge pixel.
inline bool BresenhamLine(int2 A, int2 B, const Ctx& h) {
	const uint8_t* edge = h.edge.data();
	int height = h.height, width = h.width;
	int max_step = (int)(ORA_MAX(height, width) / 30.0);
	int x0 = B.x, y0 = B.y, x1 = A.x, y1 = A.y;
	int ABx = A.x - B.x, ABy = A.y - B.y;
	if (ABx * ABx + ABy * ABy > 9 * max_step * max_step) return false;
	if (edge[x0 + y0 * width] || edge[x1 + y1 * width]) return false;
	int dx = std::abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
	int dy = std::abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
	int erro = (dx > dy ? dx : dy) / 2;
	int step = 0;
	bool tagx = true, tagy = true;
	while (tagx || tagy) {
		if (x0 == x1) tagx = false;
		if (y0 == y1) tagy = false;
		int e2 = erro;
		if (e2 > -dx) { erro -= dy; x0 += sx; }
		if (e2 < dy) { erro += dx; y0 += sy; }
		// the walk can step one pixel past the end point and so past the image border: the
		// reference reads out of bounds there; defined here as "no edge outside the image".
		if (x0 >= 0 && x0 < width && y0 >= 0 && y0 < height && edge[x0 + y0 * width]) return true;
		step += 1;
		if (step >= max_step) break;
	}
	return false;
}
inline bool BresenhamLine(short2 A, short2 B, const Ctx& h) {
	return BresenhamLine(make_int2(A.x, A.y), make_int2(B.x, B.y), h);
}

// ---- homography (APD.cu:679-748) --------------------------------------------------------------
inline void ComputeHomography(const Camera& ref_camera, const Camera& src_camera, const float4 pl, float* H) {
	float ref_C[3], src_C[3];
	ref_C[0] = -(ref_camera.R[0] * ref_camera.t[0] + ref_camera.R[3] * ref_camera.t[1] + ref_camera.R[6] * ref_camera.t[2]);
	ref_C[1] = -(ref_camera.R[1] * ref_camera.t[0] + ref_camera.R[4] * ref_camera.t[1] + ref_camera.R[7] * ref_camera.t[2]);
	ref_C[2] = -(ref_camera.R[2] * ref_camera.t[0] + ref_camera.R[5] * ref_camera.t[1] + ref_camera.R[8] * ref_camera.t[2]);
	src_C[0] = -(src_camera.R[0] * src_camera.t[0] + src_camera.R[3] * src_camera.t[1] + src_camera.R[6] * src_camera.t[2]);
	src_C[1] = -(src_camera.R[1] * src_camera.t[0] + src_camera.R[4] * src_camera.t[1] + src_camera.R[7] * src_camera.t[2]);
	src_C[2] = -(src_camera.R[2] * src_camera.t[0] + src_camera.R[5] * src_camera.t[1] + src_camera.R[8] * src_camera.t[2]);
	float R_relative[9], C_relative[3], t_relative[3];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			R_relative[3 * i + j] = src_camera.R[3 * i + 0] * ref_camera.R[3 * j + 0] + src_camera.R[3 * i + 1] * ref_camera.R[3 * j + 1] + src_camera.R[3 * i + 2] * ref_camera.R[3 * j + 2];
	for (int i = 0; i < 3; ++i) C_relative[i] = (ref_C[i] - src_C[i]);
	for (int i = 0; i < 3; ++i)
		t_relative[i] = src_camera.R[3 * i + 0] * C_relative[0] + src_camera.R[3 * i + 1] * C_relative[1] + src_camera.R[3 * i + 2] * C_relative[2];
	// Divisions by plane.w / K[0] / K[4]: evaluated as multiplication by the correctly rounded
	// reciprocal of the divisor, one reciprocal per distinct divisor.  That is how the reference is
	// actually compiled: nvcc --use_fast_math (CMakeLists.txt:21) implies -prec-div=false, i.e.
	// a/b -> a * rcp(b) (div.approx.f32).  See the numerics contract in ora_common.h.
	float tmp[9];
	if (literal_mode()) {   // APD.cu:709-730 as written: a * b / c left to right, true divisions
		for (int i = 0; i < 3; ++i) {
			H[3 * i + 0] = R_relative[3 * i + 0] - t_relative[i] * pl.x / pl.w;
			H[3 * i + 1] = R_relative[3 * i + 1] - t_relative[i] * pl.y / pl.w;
			H[3 * i + 2] = R_relative[3 * i + 2] - t_relative[i] * pl.z / pl.w;
		}
		for (int i = 0; i < 3; ++i) {
			tmp[3 * i + 0] = H[3 * i + 0] / ref_camera.K[0];
			tmp[3 * i + 1] = H[3 * i + 1] / ref_camera.K[4];
			tmp[3 * i + 2] = -H[3 * i + 0] * ref_camera.K[2] / ref_camera.K[0] - H[3 * i + 1] * ref_camera.K[5] / ref_camera.K[4] + H[3 * i + 2];
		}
	} else {
		const float inv_w = 1.0f / pl.w;
		const float inv_k0 = 1.0f / ref_camera.K[0];
		const float inv_k4 = 1.0f / ref_camera.K[4];
		for (int i = 0; i < 3; ++i) {
			H[3 * i + 0] = R_relative[3 * i + 0] - t_relative[i] * pl.x * inv_w;
			H[3 * i + 1] = R_relative[3 * i + 1] - t_relative[i] * pl.y * inv_w;
			H[3 * i + 2] = R_relative[3 * i + 2] - t_relative[i] * pl.z * inv_w;
		}
		for (int i = 0; i < 3; ++i) {
			tmp[3 * i + 0] = H[3 * i + 0] * inv_k0;
			tmp[3 * i + 1] = H[3 * i + 1] * inv_k4;
			tmp[3 * i + 2] = -H[3 * i + 0] * ref_camera.K[2] * inv_k0 - H[3 * i + 1] * ref_camera.K[5] * inv_k4 + H[3 * i + 2];
		}
	}
	H[0] = src_camera.K[0] * tmp[0] + src_camera.K[2] * tmp[6];
	H[1] = src_camera.K[0] * tmp[1] + src_camera.K[2] * tmp[7];
	H[2] = src_camera.K[0] * tmp[2] + src_camera.K[2] * tmp[8];
	H[3] = src_camera.K[4] * tmp[3] + src_camera.K[5] * tmp[6];
	H[4] = src_camera.K[4] * tmp[4] + src_camera.K[5] * tmp[7];
	H[5] = src_camera.K[4] * tmp[5] + src_camera.K[5] * tmp[8];
	H[6] = src_camera.K[8] * tmp[6];
	H[7] = src_camera.K[8] * tmp[7];
	H[8] = src_camera.K[8] * tmp[8];
}
inline float2 ComputeCorrespondingPoint(const float* H, const int2 p) {   // APD.cu:741-748
	float3 pt;
	pt.x = H[0] * p.x + H[1] * p.y + H[2];
	pt.y = H[3] * p.x + H[4] * p.y + H[5];
	pt.z = H[6] * p.x + H[7] * p.y + H[8];
	if (literal_mode()) return make_float2(pt.x / pt.z, pt.y / pt.z);
	const float inv_z = 1.0f / pt.z;   // x/z, y/z as x*rcp(z), y*rcp(z): see ComputeHomography
	return make_float2(pt.x * inv_z, pt.y * inv_z);
}

// ---- bilateral weights (APD.cu:776-788) -------------------------------------------------------
inline float ComputeBilateralWeight(const float x_dist, const float y_dist, const float pix, const float center_pix, const float sigma_spatial, const float sigma_color) {
	const float spatial_dist = sqrtf(x_dist * x_dist + y_dist * y_dist);
	const float color_dist = fabsf(pix - center_pix);
	return dvp_expf(-spatial_dist / (2.0f * sigma_spatial * sigma_spatial) - color_dist / (2.0f * sigma_color * sigma_color));
}
inline float ComputeBilateralWeight_YZL(const float, const float, const float pix, const float center_pix, const float, const float sigma_color) {
	const float color_dist = fabsf(pix - center_pix);
	return dvp_expf(-color_dist / (2.0f * sigma_color * sigma_color));
}

inline short2 GetNeighbourPoint(const int2 p, const int index, const Ctx& h) {   // APD.cu:770-774
	const unsigned offset = h.neighbours_map[p.x + p.y * h.width] * NEIGHBOUR_NUM;
	return h.neighbours[offset + index];
}

float ComputeBilateralNCCOld(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h);
float ComputeBilateralNCCNew(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h);
float ComputeGeomConsistencyCost(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h);

}  // namespace ora
#endif
