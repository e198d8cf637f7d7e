// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).
//
// CPU restatement of the DVP-MVS PatchMatch hot path (APD::RunPatchMatch and the 16 kernels it
// launches, /root/reference/APD.cu:4406-4532).  Plain C++17, no GPU, no third-party deps.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (dvp-mvs_amd/) never includes, links or calls anything in oracle/.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, and
// it cannot be built in this image (needs the CUDA toolkit incl. cuRAND and the texture unit,
// OpenCV and Boost — all absent; writing stand-ins for them is not allowed).  So this
// restatement is anchored on the reference's source expressions, cited file:line per function,
// not on outputs of the reference binary.
//
// Numerics contract (shared, by specification, with the HIP engine — DESIGN.md §Numerics).  The
// reference was built with nvcc --use_fast_math (CMakeLists.txt:21) for NVIDIA's texture unit and
// cuRAND: its low-order bits cannot be reproduced, so the contract fixes an evaluation order and a
// rounding for every expression; each item moves results at the ulp level only.
//  * every expression is evaluated in IEEE-754 binary32 (binary64 where the reference source
//    promotes to double), one rounding per operator, no FMA contraction (build with
//    -ffp-contract=off), fused operations written as explicit fmaf(); division and sqrt are
//    correctly rounded.
//  * mirroring nvcc's lowering under --use_fast_math (-prec-div=false: a/b = a*rcp(b)): the
//    divisions of ComputeHomography (by plane.w, K[0], K[4]) and ComputeCorrespondingPoint (by the
//    projective z) are evaluated as a * (1.0f / b) with a correctly rounded reciprocal, one
//    reciprocal per distinct divisor.  Every other division is a correctly rounded a / b.
//  * the NCC tap loops (APD.cu:1059-1089, 905-935) keep the source's structure (six partial sums of six
//    taps, added to the totals in order) on the TRANSPOSED walk: row by row (y offset outer, x offset inner;
//    the source walks x outer, y inner) — see ora_cost.cpp for the measured reason; the projective divide of a row is taken six taps
//    at a time through ONE division (batch_rcp, ora_core.h); the three source-side moments use one
//    fmaf per tap.  ctx.numerics = 1 switches these loops (and the sampler) to the literal
//    per-operator evaluation in the reference's own order; tests/test_oracle_kat.py measures the
//    distance between the two (NCC costs: 4e-6 median, 5e-4 max).
//  * exp() is the polynomial dvp_expf below (<= 1 ulp on the ranges used).
//  * rsqrtf(x) is restated as 1.0f / sqrtf(x).
//  * cuRAND XORWOW seeded by clock64() (APD.cu:1270) is replaced by a counter-based generator
//    keyed by (seed, pixel, site, k): every logical sampling site owns a sub-stream, so the
//    number of draws one site consumes never shifts another site's values.
//  * tex2D<float>(x + 0.5f, y + 0.5f) with cudaFilterModeLinear / unnormalised coords / clamp
//    (APD.cpp:1501-1517) is restated in software as a function of the pixel coordinate (x, y): the
//    unit samples at coordinate - 0.5, which cancels the call sites' + 0.5f.  Clamp-to-edge;
//    sampler 0 ("cuda8") converts the coordinate to fixed point with 8 fractional bits, round half
//    up (CUDA Programming Guide, "Linear Filtering"); sampler 1 uses the exact floor / fraction.
//  * MIN/MAX are OpenCV's macros (opencv2/core/cvdef.h): MIN(a,b)=((a)>(b)?(b):(a)),
//    MAX(a,b)=((a)<(b)?(b):(a)); float min/max/fminf/fmaxf in device code are CUDA's
//    NaN-ignoring fminf/fmaxf.
#ifndef ORA_COMMON_H_
#define ORA_COMMON_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>

namespace ora {

// numerics == 1 ("literal", Oracle.set_numerics): every site the numerics contract restates is evaluated
// as the reference's SOURCE TEXT says instead — true divisions in ComputeHomography /
// ComputeCorrespondingPoint, x-offset-outer tap order with per-outer-index partial moment sums
// (APD.cu:1059-1089, 904-926), one division per tap, tex2D(x + 0.5f), libm expf, the
// `complex` sigmoid in double.  Process-wide switch (the free functions below take no context); used by
// tests only, to measure how far the contract is from a literal reading (tests/test_literal_mode.py).
inline int& literal_mode() { static int v = 0; return v; }

struct float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
struct float2 { float x, y; };
struct int2 { int x, y; };
struct short2 { short x, y; };

inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline short2 make_short2(int x, int y) { return short2{(short)x, (short)y}; }

#define ORA_MIN(a, b) ((a) > (b) ? (b) : (a))
#define ORA_MAX(a, b) ((a) < (b) ? (b) : (a))

// main.h:39-49
constexpr int MAX_IMAGES = 32;
constexpr int NEIGHBOUR_NUM = 12;
constexpr int EDGE_NEIGH_NUM = 8;
constexpr int LAB_BOUNDARY_NUM = 8;
constexpr int MAX_SEARCH_RADIUS = 4096;

// main.h:58-67 (112 bytes)
struct Camera {
	float K[9];
	float R[9];
	float t[3];
	float c[3];
	int height;
	int width;
	float depth_min;
	float depth_max;
};
static_assert(sizeof(Camera) == 112, "Camera layout");

enum RunState { FIRST_INIT = 0, REFINE_INIT = 1, REFINE_ITER = 2 };  // main.h:74-78
enum PixelState { WEAK = 0, STRONG = 1, UNKNOWN = 2 };                // main.h:80-84

// main.h:86-112 (76 bytes; bool == 1 byte)
struct PatchMatchParams {
	int max_iterations;
	int num_images;
	float sigma_spatial;
	float sigma_color;
	int top_k;
	float depth_min;
	float depth_max;
	uint8_t geom_consistency;
	int strong_radius;
	int strong_increment;
	int weak_radius;
	int weak_increment;
	uint8_t use_APD;
	uint8_t use_edge;
	uint8_t use_limit;
	uint8_t use_label;
	uint8_t use_detail;
	uint8_t use_radius;
	int weak_peak_radius;
	int rotate_time;
	float ransac_threshold;
	float geom_factor;
	int state;
};
static_assert(sizeof(PatchMatchParams) == 76, "PatchMatchParams layout");

// ---------------------------------------------------------------------------------------------
// exp: Cephes-style expf.  n = rint(x*log2e); r = x - n*ln2 (two-term); degree-6 polynomial;
// scale by 2^n through the exponent field (two-step so that results down to the denormal range
// stay correct).  Only fmaf/mul/add/rint: bit-identical on any IEEE-754 machine.
inline float dvp_expf_contract(float x) {
	if (!(x > -103.0f)) return (x != x) ? x : 0.0f;   // underflow (NaN propagates)
	if (x > 88.72f) return INFINITY;
	const float n = rintf(x * 1.44269504088896341f);
	float r = fmaf(n, -0.693359375f, x);
	r = fmaf(n, 2.12194440e-4f, r);
	float p = 1.9875691500e-4f;
	p = fmaf(p, r, 1.3981999507e-3f);
	p = fmaf(p, r, 8.3334519073e-3f);
	p = fmaf(p, r, 4.1665795894e-2f);
	p = fmaf(p, r, 1.6666665459e-1f);
	p = fmaf(p, r, 5.0000001201e-1f);
	const float r2 = r * r;
	float y = fmaf(p, r2, r) + 1.0f;
	int ni = (int)n;
	// split the scaling so 2^n never overflows/underflows the exponent field on its own
	int n1 = ni / 2, n2 = ni - n1;
	uint32_t b1 = (uint32_t)(n1 + 127) << 23, b2 = (uint32_t)(n2 + 127) << 23;
	float s1, s2;
	std::memcpy(&s1, &b1, 4);
	std::memcpy(&s2, &b2, 4);
	return (y * s1) * s2;
}
// exp() at the reference's call sites: the contract polynomial, or libm's expf in literal mode
inline float dvp_expf(float x) { return literal_mode() ? expf(x) : dvp_expf_contract(x); }

// ---------------------------------------------------------------------------------------------
// Counter-based RNG (replaces curandState, APD.cu:1258-1271).  splitmix64 finaliser over a
// 64-bit word built from (seed, pixel, site, k).  site = (phase << 16) | (iter << 8) | sub.
inline uint32_t dvp_rand_u32(uint64_t seed, uint32_t pixel, uint32_t site, uint32_t k) {
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)pixel + ((uint64_t)site << 32));
	z += 0xD1B54A32D192ED03ull * (uint64_t)(k + 1u);
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27; z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return (uint32_t)(z >> 32);
}
// curand_uniform: (0, 1]
inline float dvp_rand_uniform(uint64_t seed, uint32_t pixel, uint32_t site, uint32_t k) {
	return (float)((dvp_rand_u32(seed, pixel, site, k) >> 8) + 1u) * (1.0f / 16777216.0f);
}

enum RngPhase {
	PH_RANDOM_INIT = 1,
	PH_STRONG = 2,
	PH_RANSAC = 3,
	PH_WEAK = 4,
	PH_NEIGHBOURS = 5,
};
enum RngSub {
	SUB_VIEW = 0,        // 15 view-selection draws
	SUB_DEPTH_RAND = 1,  // depth_rand
	SUB_NORMAL = 2,      // GenerateRandomNormal_YZL rejection stream
	SUB_DEPTH_PERT = 3,  // depth_perturbed
	SUB_LIMIT = 4,       // edge_limit coin (GenNeighbours / RANSACToGetFitPlane)
	SUB_SEARCH = 5,      // GenNeighbours directional search shifts
	SUB_RANSAC = 6,      // RANSAC index triplets
};
inline uint32_t rng_site(int phase, int iter, int sub) {
	return ((uint32_t)phase << 16) | ((uint32_t)(iter & 0xff) << 8) | (uint32_t)sub;
}

struct Rng {
	uint64_t seed;
	uint32_t pixel, site, k;
	Rng(uint64_t s, uint32_t p, uint32_t st) : seed(s), pixel(p), site(st), k(0) {}
	uint32_t next() { return dvp_rand_u32(seed, pixel, site, k++); }
	float uniform() { return dvp_rand_uniform(seed, pixel, site, k++); }
};

}  // namespace ora
#endif
