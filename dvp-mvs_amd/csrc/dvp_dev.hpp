// dvp_dev.hpp — device-side types, numerics primitives and the software sampler of the MI355X
// PatchMatch engine.  Everything here is plain C++ over IEEE-754 binary32 (explicit fmaf where a
// fused operation is wanted; the build uses -ffp-contract=off) so that results are bit-stable
// and can be checked against the CPU oracle bit for bit.
//
// DVP_HD marks functions that run on the device.  tests/emul builds the same headers with g++
// (DVP_HD empty) to exercise kernel logic on machines without a GPU; the shipped library never
// contains or falls back to that build.
#ifndef DVP_DEV_HPP_
#define DVP_DEV_HPP_

#include <stdint.h>
#include <math.h>
#include <float.h>
#include <string.h>
#include "../../include/dvp_mvs.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DVP_HD __host__ __device__ __forceinline__
#define DVP_HD_NOINLINE __host__ __device__ __noinline__
#else
#define DVP_HD inline
#define DVP_HD_NOINLINE inline
#endif
// Streaming hints for records that are written once and read once by a later launch (the sweep passes' and the strong update's cost
// records): non-temporal stores / loads, so that they do not push the image rows the evaluators gather out of L2.  Same values either way.
// DVP_STREAM_HINTS bits: 1 sweep-cost stores, 2 sweep-cost loads, 4 slot-cost stores, 8 slot-cost loads.
#ifndef DVP_STREAM_HINTS
#define DVP_STREAM_HINTS 3   // sweep-cost records: depth_to_weak site 521.4 -> 517.1 ms at cfg3; the slot-cost records (12) lose 42 ms: the decision launch re-reads their lines through L2 (profiles/r06_ab_notes.txt)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define DVP_NT_STORE(bit, p, v) do { if ((DVP_STREAM_HINTS) & (bit)) __builtin_nontemporal_store((v), (p)); else *(p) = (v); } while (0)
#define DVP_NT_LOAD(bit, p) (((DVP_STREAM_HINTS) & (bit)) ? __builtin_nontemporal_load(p) : *(p))
#else
#define DVP_NT_STORE(bit, p, v) (*(p) = (v))
#define DVP_NT_LOAD(bit, p) (*(p))
#endif

namespace dvp {

struct f4 { float x, y, z, w; };
struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct i2 { int x, y; };
struct s2 { short x, y; };

DVP_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
DVP_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DVP_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
DVP_HD i2 mki2(int x, int y) { i2 r; r.x = x; r.y = y; return r; }
DVP_HD s2 mks2(int x, int y) { s2 r; r.x = (short)x; r.y = (short)y; return r; }

// Per source view constants of the plane-induced homography (camera-only sub-expressions of
// ComputeHomography, APD.cu:681-707, evaluated once per view in the same operation order).
struct ViewConst {
	float Rrel[9];
	float trel[3];
	// everything else ComputeHomography and the centre test read from the two cameras, so that an
	// evaluation fetches ONE 96-byte record of wave-uniform data (scalar loads, see load_view)
	float sK0, sK2, sK4, sK5, sK8;   // source K[0], K[2], K[4], K[5], K[8]
	float rK2, rK5;                   // reference K[2], K[5]
	float inv_k0, inv_k4;             // 1.0f / reference K[0], K[4] (one correctly rounded reciprocal per divisor)
	float fw, fh;                     // source width / height as float (the centre test compares floats)
	float pad;
};

// The buffer bundle every kernel receives (the engine's DataPassHelper, APD.h:60-92).
struct AnchorRec;   // dvp_weak_wave.hpp
struct WeakRec;     // dvp_weak_phased.hpp
struct Dev {
	int width, height, num_images;
	int pitch;                 // floats per padded image row (multiple of 64 -> 256 B aligned rows)
	int org;                   // element offset of pixel (0,0) inside a padded plane (= PAD*pitch + PAD)
	size_t plane_stride;       // floats per padded depth plane ((H + 2*PAD) * pitch); an image plane is 2x that (row pairs)
	int sampler;               // 0 = 8-bit interpolation weights, 1 = exact
	int weak_count;
	uint64_t seed;
	DvpParams params;
	// uniform constants derived from params on the host (GenNeighbours, APD.cu:3375-3380)
	float nb_cos, nb_sin, nb_thresh;
	int nb_shift_range;
	const float* images;       // [num_images][H + 2*PAD][pitch][2] row-pair planes {I(x,y), I(x,y+1)}, border replicated (== clamp addressing)
	// the same images as BYTES (uchar2 {I(x,y), I(x,y+1)}) in 128-byte TILES, or null: kept only when every texel of
	// every image is an integer in [0, 255], i.e. images decoded from 8-bit files and used at their native size
	// (APD.cpp:1057-1069, scale_size == 1).  A bilinear footprint is then 4 bytes instead of 16.  A tile is 8 rows of
	// 8 elements = 7 pixels + the first pixel of the right-hand neighbour again (so that every footprint x, x+1 lies
	// inside one tile row): the 9 taps of an anchor sub-patch — arbitrary offsets in an 11x11 window — land in 3.7 tiles
	// on average instead of 6.6 lines of a row-major plane (they rarely share an image row).  The gather-bound weak
	// update reads these (same values, so same results).
	const uint8_t* images8;
	int img8_tiles_x;            // tiles per tile row
	size_t img8_plane_bytes;     // bytes of one image
	const float* depths;       // same layout (geom_consistency only)
	const DvpCamera* cameras;  // [num_images]
	const ViewConst* views;    // [num_images] (index 0 unused)
	// window offsets of GenEdgeInform by 30-degree sector (APD.cu:797-821), r = weak_radius, host-built:
	// sector s owns sector_taps[sector_start[s] .. sector_start[s+1]) in visit order (i outer, j inner);
	// an entry is (i + r) | (j + r) << 16
	const int* sector_taps;
	const int* sector_start;   // [13]
	f4* planes;
	const f4* planes_snap;     // pre-launch copy for the strong update (direction-4 same-colour reads)
	float* costs;
	const float* costs_snap;
	uint32_t* selected_views;  // +width zeroed tail (APD.cu:2473 reads one row past the end)
	uint8_t* view_weight;      // 32 per pixel
	uint8_t* weak_info;
	uint8_t* weak_reliable;
	s2* weak_nearest_strong;
	const int* neighbours_map;
	s2* neighbours;            // 12 per WEAK pixel
	s2* gn_points;             // GenNeighbours hand-over: the 32 directional slots per WEAK pixel (holes = (-1,-1))
	int* gn_count;             // ... and how many of them are filled
	f4* fit_planes;
	s2* candidate;             // [view][pixel][8]: cand_ptr(); 64 x-adjacent pixels write/read 2 KB contiguous per view
	AnchorRec* anchor_tab;     // [WEAK pixel][view][11]: reference side of the anchor sub-patches, built once per pass (dvp_weak_wave.hpp); null: the weak update forms it per item
	// the weak update as seven launches (dvp_weak_phased.hpp), per WEAK pixel, or null (the one-wave form):
	WeakRec* weak_rec;         // hand-over record between the launches
	f2* weak_ctab;             // [36] centre-patch table (w, w * ref) of the update
	float* weak_ev;            // [8][S] cost vectors of the planes an evaluation launch took
	const uint8_t* edge;
	int* search_pos;           // [16][L]: sample positions of the strong update's 16 propagation slots (strong_search_px)
	// split strong update (dvp_strong_eval / _decide / _refine): the cost vectors of the 16 propagation slots + the current plane,
	// [slot][view][row * half_w + x / 2] (a red/black launch touches every second pixel of a row), or null
	float* slot_costs;         // [half_w * H][17][S] (dvp_strong.hpp: slot_cost_index)
	float* strong_rec;         // [SR_FIELDS][half_w * H]: hand-over from dvp_strong_decide to dvp_strong_refine
	// DepthToWeak + LocalRefine as view-compacted passes (dvp_strong.hpp: sweep_*), or null (the fused per-pixel kernel)
	f4* sweep_rec;             // [2][L]: (camera-frame normal, depth) and (mean baseline, disparity, weight sum, flags) per pixel
	float* sweep_cost;         // [L/64][S][kSweepFields][64] (sweep_cost_index): per (view, sweep slot, pixel) costs written by dvp_sweep_eval
	float* sweep_pc;           // [61][L]: the folded cost line of the central window, handed from the first decision pass to the second
	int sweep_px0;             // sweep passes in bands of rows (a context told to keep sweep_cost small): sweep_cost holds the pixels from this linear index on,
	int sweep_row0, sweep_row1;   // and the decision passes of a band take the rows [row0, row1) (row1 == 0: the whole image)
	int half_w;
	uint32_t* edge_bits;       // the edge map as 32x32-pixel bit tiles (128 B each), see edge_bit()
	int edge_tiles_x;
	uint32_t* strong_bits;     // same tiling, bit = (weak_info == STRONG); valid during FindNearestStrongPoint / GenNeighbours
	uint32_t* strong_bits_t;   // the same map with the tiles transposed: word c of a tile = its column c, bit r = row r (column segments in 1-2 words)
	// summed-area table of edge-pixel counts over 8x8-pixel cells, (cells_y + 1) x (cells_x + 1) ints, first
	// row / column zero: edge_sat[(cy + 1) * (cells_x + 1) + cx + 1] = #edge pixels in cells [0..cx] x [0..cy].
	// Lets a line test prove "no edge pixel anywhere near this segment" with four loads.
	int* edge_sat;
	int sat_cells_x, sat_cells_y;
	s2* edge_neigh;            // 8 per pixel
	const int* label;
	s2* label_boundary;        // 8 per WEAK pixel
	s2* label_stop;            // 8 per pixel: nearest pixel with label -1 per direction (valid during GenEdgeInform when use_label)
	float* complex_;           // per WEAK pixel
	int* radius;
	// WEAK pixels compacted: pixel indices in raster order, black ((x+y) even) first, then red;
	// the weak-path kernels launch one lane per list entry instead of one lane per image pixel
	const int* weak_list;
	int weak_black, weak_red;
	unsigned long long* eval_counter;   // profiling builds only (may be null)
};

#define DVP_MIN(a, b) ((a) > (b) ? (b) : (a))   // OpenCV cvdef.h semantics (used by the reference)
#define DVP_MAX(a, b) ((a) < (b) ? (b) : (a))

// ---- exp (specification shared with the oracle: Cephes-style, fmaf only) ---------------------
DVP_HD float dvp_expf(float x) {
	if (!(x > -103.0f)) return (x != x) ? x : 0.0f;
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
	const float y = fmaf(p, r2, r) + 1.0f;
	const int ni = (int)n;
	const int n1 = ni / 2, n2 = ni - n1;
	union { uint32_t u; float f; } s1, s2;
	s1.u = (uint32_t)(n1 + 127) << 23;
	s2.u = (uint32_t)(n2 + 127) << 23;
	return (y * s1.f) * s2.f;
}

// ---- counter-based RNG (replaces curandState; specification shared with the oracle) ----------
DVP_HD uint32_t rand_u32(uint64_t seed, uint32_t pixel, uint32_t site, uint32_t k) {
	uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)pixel + ((uint64_t)site << 32));
	z += 0xD1B54A32D192ED03ull * (uint64_t)(k + 1u);
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27; z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return (uint32_t)(z >> 32);
}
DVP_HD float rand_uniform(uint64_t seed, uint32_t pixel, uint32_t site, uint32_t k) {   // (0, 1]
	return (float)((rand_u32(seed, pixel, site, k) >> 8) + 1u) * (1.0f / 16777216.0f);
}
enum { PH_RANDOM_INIT = 1, PH_STRONG = 2, PH_RANSAC = 3, PH_WEAK = 4, PH_NEIGHBOURS = 5 };
enum { SUB_VIEW = 0, SUB_DEPTH_RAND = 1, SUB_NORMAL = 2, SUB_DEPTH_PERT = 3, SUB_LIMIT = 4, SUB_SEARCH = 5, SUB_RANSAC = 6 };
DVP_HD uint32_t rng_site(int phase, int iter, int sub) {
	return ((uint32_t)phase << 16) | ((uint32_t)(iter & 0xff) << 8) | (uint32_t)sub;
}
struct Rng {
	uint64_t seed;
	uint32_t pixel, site, k;
	DVP_HD Rng(uint64_t s, uint32_t p, uint32_t st) : seed(s), pixel(p), site(st), k(0) {}
	DVP_HD uint32_t next() { return rand_u32(seed, pixel, site, k++); }
	DVP_HD float uniform() { return rand_uniform(seed, pixel, site, k++); }
};

// ---- software texture unit (gfx950 has no tex2D path; semantics of APD.cpp:1501-1517) --------
// Image planes carry kImgPad replicated border pixels on every side: reading the padded plane at
// an unclamped coordinate in [-kImgPad, W-1+kImgPad] IS clamp-to-edge addressing, so the bilinear
// footprint needs no integer clamps.
constexpr int kImgPad = 2;

// the 8 visibility-prior offsets of (pixel, source view v = 0..S-1).  Host-visible layout of
// DVP_BUF_CANDIDATE stays [pixel][view][8] (dvp_download_buffer transposes).
DVP_HD size_t cand_index(const Dev& d, int pixel, int v) { return ((size_t)v * ((size_t)d.width * d.height) + (size_t)pixel) * 8; }

DVP_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tex2D(map, ix + 0.5f, iy + 0.5f): exact texel, clamp-to-edge (arbitrary integer coordinates).
// Plain planes (the depth maps of the geometric-consistency term).
DVP_HD float tex_texel(const float* img, int org, int pitch, int W, int H, int ix, int iy) {
	return img[org + clampi(iy, 0, H - 1) * pitch + clampi(ix, 0, W - 1)];
}

// Image planes are stored as ROW PAIRS: element (x, y) of a plane is the float2 {I(x,y), I(x,y+1)}.
// The 2x2 bilinear footprint {I(x,y), I(x,y+1), I(x+1,y), I(x+1,y+1)} is then 16 contiguous bytes:
// ONE global_load_dwordx4 per tap instead of two dwordx2 loads on two image rows, i.e. half the
// load instructions and — for lanes whose hypotheses are unrelated (random draws, propagated planes
// of far-away pixels, anchor sub-patches) — half the cache lines the texture path has to look up.
// Costs 2x the image bytes in HBM (each texel is stored in its own row pair and in the one above).
DVP_HD float img_texel(const float* img, int org, int pitch, int W, int H, int ix, int iy) {
	return img[(size_t)(org + clampi(iy, 0, H - 1) * pitch + clampi(ix, 0, W - 1)) * 2];
}

// four adjacent floats at an 8-byte aligned address: one global_load_dwordx4 with a 32-bit byte
// offset from a wave-uniform base (SGPR base + VGPR offset addressing)
DVP_HD void load_quad(const float* base, unsigned byte_off, float* a, float* b, float* c, float* e) {
#if defined(__HIPCC__)
	typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
	const f4u t = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(base) + byte_off);
	*a = t.x; *b = t.y; *c = t.z; *e = t.w;
#else
	const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
	*a = p[0]; *b = p[1]; *c = p[2]; *e = p[3];
#endif
}
// Image format of a kernel instantiation: FMT 0 = float row pairs (Dev::images), FMT 1 = tiled byte pairs
// (Dev::images8).
#ifndef DVP_IMG8_TW
#define DVP_IMG8_TW 7
#endif
// DVP_IMG8_TW = 7 / 15: uchar2 elements, tile rows of TW pixels + one repeated; 8: QUAD elements (uchar4
// {I(x,y), I(x,y+1), I(x+1,y), I(x+1,y+1)}, nothing repeated, 8 x 4 pixels per tile, power-of-two addressing)
constexpr bool kT8Quad = DVP_IMG8_TW == 8;
constexpr int kT8W = DVP_IMG8_TW;                            // unique pixels per tile row
constexpr int kT8E = kT8Quad ? 8 : kT8W + 1;                 // elements per tile row
constexpr int kT8B = kT8Quad ? 4 : 2;                        // bytes per element
constexpr int kT8H = 128 / (kT8E * kT8B);                    // tile rows
constexpr unsigned kT8Mul = kT8W == 7 ? 74899u : (kT8W == 15 ? 34953u : 65536u);   // x / kT8W == (x * kT8Mul) >> 19 for x < 70000
DVP_HD int img8_tiles_x(int W) { return (W + 2 * kImgPad + kT8W - 1) / kT8W; }
DVP_HD int img8_tiles_y(int H) { return (H + 2 * kImgPad + kT8H - 1) / kT8H; }
// byte offset of the footprint {I(i0,j0), I(i0,j0+1), I(i0+1,j0), I(i0+1,j0+1)} inside a tiled byte plane;
// i0 >= -PAD, j0 >= -PAD (pixel coordinates; the PAD frame is part of the plane)
DVP_HD unsigned img8_offset(int tiles_x, int i0, int j0) {
	const unsigned px = (unsigned)(i0 + kImgPad), py = (unsigned)(j0 + kImgPad);
	// px < 2^15 + PAD (dvp_ctx_create), kT8Mul < 2^17: the products fit 32 bits and both factors 24 — v_mul_u32_u24 (full rate)
	// instead of v_mul_lo_u32 (quarter rate).  Round 3 measured no gain from this while the weak update waited for its gathers;
	// its propagation launch now is VALU-bound (PMC r05: 0.56 of the issue peak, gather roof 0.35)
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned tx = __umul24(px, kT8Mul) >> 19;   // px / kT8W
	const unsigned ex = px - __umul24((unsigned)kT8W, tx);
	const unsigned ty = py / (unsigned)kT8H, ey = py % (unsigned)kT8H;
	const unsigned tile = __umul24(ty, (unsigned)tiles_x) + tx;
#else
	const unsigned tx = (px * kT8Mul) >> 19;   // px / kT8W
	const unsigned ex = px - (unsigned)kT8W * tx;
	const unsigned ty = py / (unsigned)kT8H, ey = py % (unsigned)kT8H;
	const unsigned tile = ty * (unsigned)tiles_x + tx;
#endif
	return (tile << 7) + ey * (unsigned)(kT8B * kT8E) + ex * (unsigned)kT8B;
}
template <int FMT> DVP_HD const void* img_plane(const Dev& d, int v);
template <> DVP_HD const void* img_plane<0>(const Dev& d, int v) { return d.images + (size_t)v * d.plane_stride * 2; }
template <> DVP_HD const void* img_plane<1>(const Dev& d, int v) { return d.images8 + (size_t)v * d.img8_plane_bytes; }
// `off`: byte offset of the footprint inside the plane of the format (tex_coord_t)
template <int FMT> DVP_HD void load_quad_t(const void* base, unsigned off, float* a, float* b, float* c, float* e);
template <> DVP_HD void load_quad_t<0>(const void* base, unsigned off, float* a, float* b, float* c, float* e) {
	load_quad(static_cast<const float*>(base), off, a, b, c, e);
}
template <> DVP_HD void load_quad_t<1>(const void* base, unsigned off, float* a, float* b, float* c, float* e) {
	// one 4-byte load at a 2-byte aligned address, four v_cvt_f32_ubyteN
	uint32_t t;
#if defined(__HIP_DEVICE_COMPILE__)
	typedef uint32_t u32_a2 __attribute__((aligned(2)));
	t = *reinterpret_cast<const u32_a2*>(static_cast<const char*>(base) + off);
#else
	memcpy(&t, static_cast<const char*>(base) + off, 4);
#endif
	*a = (float)(t & 255u); *b = (float)((t >> 8) & 255u); *c = (float)((t >> 16) & 255u); *e = (float)(t >> 24);
}
// texel (ix, iy) of the reference image (plane 0), clamp-to-edge
template <int FMT> DVP_HD float ref_texel_t(const Dev& d, int ix, int iy);
template <> DVP_HD float ref_texel_t<0>(const Dev& d, int ix, int iy) { return img_texel(d.images, d.org, d.pitch, d.width, d.height, ix, iy); }
template <> DVP_HD float ref_texel_t<1>(const Dev& d, int ix, int iy) {
	return (float)d.images8[img8_offset(d.img8_tiles_x, clampi(ix, 0, d.width - 1), clampi(iy, 0, d.height - 1))];
}
// clamp(v, lo, hi) with NaN -> lo: fminf(fmaxf(v, lo), hi).  One v_med3_f32 on the device (with a
// NaN operand the instruction returns min3 of the other two == lo).
DVP_HD float clampf_nan_lo(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
	return fminf(fmaxf(v, lo), hi);
#endif
}
// keeps the instruction scheduler from moving anything across this point (no code is emitted)
DVP_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
	__builtin_amdgcn_sched_barrier(0);
#endif
}
// (int)floorf(v) for |v| < 2^31: v_cvt_flr_i32_f32
DVP_HD int floor_to_int(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
	int r;
	asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
	return r;
#else
	return (int)floorf(v);
#endif
}

// The reference's tex2D(img, x + 0.5f, y + 0.5f) with cudaFilterModeLinear, unnormalised
// coordinates, clamp addressing (APD.cpp:1501-1517), as a function of the PIXEL coordinate (x, y):
// the texture unit samples at (coordinate - 0.5), so the +0.5f the callers add cancels (DESIGN.md
// §Numerics: the float rounding of that add/subtract pair is not reproduced).
//   SMP 0 ("cuda8"): the coordinate is converted to fixed point with 8 fractional bits, round
//          half up, as the texture unit does; texel index = integer part, weights = fraction/256.
//   SMP 1: exact floor / fraction in binary32.
// Three pieces so that a caller can issue many fetches before consuming any of them:
//   tex_coord : coordinate -> byte offset of the footprint + the two interpolation weights
//   load_quad : the 16-byte fetch
//   tex_lerp  : the bilinear blend
// interpolation weights of one footprint as they wait for the fetch: SMP 0 keeps the two 8-bit
// fractions packed in one register, SMP 1 the two exact fractions
template <int SMP> struct TapW;
template <> struct TapW<0> { unsigned pk; };
template <> struct TapW<1> { float a, b; };
DVP_HD void tap_weights(const TapW<0>& w, float* a, float* b) {
	*a = (float)(w.pk & 255u) * (1.0f / 256.0f);          // v_cvt_f32_ubyte0
	*b = (float)((w.pk >> 8) & 255u) * (1.0f / 256.0f);   // v_cvt_f32_ubyte1
}
DVP_HD void tap_weights(const TapW<1>& w, float* a, float* b) { *a = w.a; *b = w.b; }

DVP_HD unsigned tex_offset(int pitch, int i0, int j0, unsigned plane_off = 0) {
	// i0 in [-1, W], j0 in [-1, H]: the footprint [i0, i0+1] x [j0, j0+1] lies inside the padded
	// plane; (j0 + PAD) * pitch + (i0 + PAD) >= 0, the PAD terms are a wave-uniform constant
	// |j0|, pitch < 2^23: 24-bit multiply (v_mad_i32_i24, full rate; a 32-bit multiply is quarter rate)
#if defined(__HIP_DEVICE_COMPILE__)
	const int e = __mul24(j0, pitch) + i0;
#else
	const int e = j0 * pitch + i0;
#endif
	return ((unsigned)e << 3) + ((unsigned)((kImgPad * pitch + kImgPad) * 8) + plane_off);   // plane_off: byte offset of the lane's image plane when the base is the whole image set (probe)
}
// coordinate -> integer footprint origin (i0, j0) + the two interpolation weights.
// CLAMP = false: the caller has PROVEN -1 <= x <= W and -1 <= y <= H (patch_stays_inside, dvp_ncc.hpp), for which the
// clamp is the identity — same result, two instructions per tap fewer.
template <bool CLAMP = true>
DVP_HD void tex_origin(int W, int H, float x, float y, int* i0, int* j0, TapW<0>* w) {
	const float xb = CLAMP ? clampf_nan_lo(x, -1.0f, (float)W) : x;
	const float yb = CLAMP ? clampf_nan_lo(y, -1.0f, (float)H) : y;
	const int qx = floor_to_int(fmaf(xb, 256.0f, 0.5f));   // in [-256, 256 W]
	const int qy = floor_to_int(fmaf(yb, 256.0f, 0.5f));
	w->pk = ((unsigned)qx & 255u) | ((unsigned)qy << 8);    // bits 8..15 = fraction of y
	*i0 = qx >> 8;
	*j0 = qy >> 8;
}
template <bool CLAMP = true>
DVP_HD void tex_origin(int W, int H, float x, float y, int* i0, int* j0, TapW<1>* w) {
	const float xb = CLAMP ? clampf_nan_lo(x, -1.0f, (float)W) : x;
	const float yb = CLAMP ? clampf_nan_lo(y, -1.0f, (float)H) : y;
	const float fx = floorf(xb), fy = floorf(yb);
	w->a = xb - fx;
	w->b = yb - fy;
	*i0 = (int)fx;
	*j0 = (int)fy;
}
template <int SMP, bool CLAMP = true>
DVP_HD void tex_coord(int pitch, int W, int H, float x, float y, unsigned* off, TapW<SMP>* w, unsigned plane_off = 0) {
	int i0, j0;
	tex_origin<CLAMP>(W, H, x, y, &i0, &j0, w);
	*off = tex_offset(pitch, i0, j0, plane_off);
}
// the same for a plane of format FMT
template <int FMT, int SMP>
DVP_HD void tex_coord_t(const Dev& d, float x, float y, unsigned* off, TapW<SMP>* w) {
	int i0, j0;
	tex_origin(d.width, d.height, x, y, &i0, &j0, w);
	*off = FMT ? img8_offset(d.img8_tiles_x, i0, j0) : tex_offset(d.pitch, i0, j0);
}
// q = {I(i,j), I(i,j+1), I(i+1,j), I(i+1,j+1)}
DVP_HD float tex_lerp(float a, float b, float t00, float t01, float t10, float t11) {
	const float top = fmaf(a, t10 - t00, t00);
	const float bot = fmaf(a, t11 - t01, t01);
	return fmaf(b, bot - top, top);
}
template <int SMP>
DVP_HD float tex_linear_t(const float* img, int pitch, int W, int H, float x, float y) {
	unsigned off;
	TapW<SMP> w;
	float a, b, t00, t01, t10, t11;
	tex_coord(pitch, W, H, x, y, &off, &w);
	load_quad(img, off, &t00, &t01, &t10, &t11);
	tap_weights(w, &a, &b);
	return tex_lerp(a, b, t00, t01, t10, t11);
}
DVP_HD float tex_linear(const float* img, int pitch, int W, int H, float x, float y, int sampler) {
	return sampler == 0 ? tex_linear_t<0>(img, pitch, W, H, x, y) : tex_linear_t<1>(img, pitch, W, H, x, y);
}

// Reciprocals of up to 6 projective denominators with ONE correctly rounded division (numerics
// contract: the projective divide of a patch row is taken six taps at a time).  Prefix products,
// 1/product, then the factors are peeled off again; every step is a plain binary32 multiply.
// 15 multiplies + 1 division per 6 taps instead of 6 divisions (an IEEE division is 11 VALU ops).
DVP_HD void batch_rcp(const float* z, int n, float* iz) {
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

// Edge map as bit tiles for the line walks of the weak path: tile (tx, ty) = 32 words, word r = row
// r of the tile, bit b = column b.  A walk of ~100 pixels in any direction touches 4-6 tiles
// (128 B each; the whole map is W*H/8 bytes and stays in L2) instead of one cache line per step.
DVP_HD int edge_tiles_x(int W) { return (W + 31) >> 5; }
// word of pixel (x, y) in a bit-tiled map; tile row * tiles_x < 2^24: a 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate)
DVP_HD int tile_word(int tiles_x, int x, int y) {
#if defined(__HIP_DEVICE_COMPILE__)
	return ((__mul24(y >> 5, tiles_x) + (x >> 5)) << 5) + (y & 31);
#else
	return (((y >> 5) * tiles_x + (x >> 5)) << 5) + (y & 31);
#endif
}
DVP_HD size_t edge_bits_words(int W, int H) { return (size_t)edge_tiles_x(W) * ((H + 31) >> 5) * 32; }
DVP_HD unsigned edge_bit(const Dev& d, int x, int y) {   // 0 <= x < W, 0 <= y < H
	const unsigned w = d.edge_bits[tile_word(d.edge_tiles_x, x, y)];
	return (w >> (x & 31)) & 1u;
}
DVP_HD unsigned strong_bit(const Dev& d, int x, int y) {
	const unsigned w = d.strong_bits[tile_word(d.edge_tiles_x, x, y)];
	return (w >> (x & 31)) & 1u;
}
DVP_HD int sat_cells(int n) { return (n + 7) >> 3; }
// edge pixels in the pixel rectangle [x0, x1] x [y0, y1] widened to whole 8x8 cells (an upper bound that is 0
// exactly when the widened rectangle holds no edge pixel); coordinates are clamped to the image
DVP_HD int edge_count_upper(const Dev& d, int x0, int y0, int x1, int y1) {
	const int cx0 = clampi(x0, 0, d.width - 1) >> 3, cx1 = clampi(x1, 0, d.width - 1) >> 3;
	const int cy0 = clampi(y0, 0, d.height - 1) >> 3, cy1 = clampi(y1, 0, d.height - 1) >> 3;
	const int P = d.sat_cells_x + 1;
	return d.edge_sat[(cy1 + 1) * P + cx1 + 1] - d.edge_sat[cy0 * P + cx1 + 1] - d.edge_sat[(cy1 + 1) * P + cx0] + d.edge_sat[cy0 * P + cx0];
}
// one word of a bit-tiled map from a byte map (host loop / one thread per word):
// bit = (byte != 0) when `equals` < 0, else (byte == equals)
DVP_HD uint32_t pack_edge_word(const uint8_t* edge, int W, int H, int tiles_x, size_t word, int equals = -1) {
	const int r = (int)(word & 31);
	const size_t tile = word >> 5;
	const int ty = (int)(tile / tiles_x), tx = (int)(tile - (size_t)ty * tiles_x);
	const int y = ty * 32 + r;
	uint32_t v = 0;
	if (y < H)
		for (int b = 0; b < 32; ++b) {
			const int x = tx * 32 + b;
			if (x < W && (equals < 0 ? edge[(size_t)y * W + x] != 0 : edge[(size_t)y * W + x] == equals)) v |= 1u << b;
		}
	return v;
}

// transposed tiles: word = column of the tile, bit = row
DVP_HD uint32_t pack_edge_word_t(const uint8_t* edge, int W, int H, int tiles_x, size_t word, int equals) {
	const int cidx = (int)(word & 31);
	const size_t tile = word >> 5;
	const int ty = (int)(tile / tiles_x), tx = (int)(tile - (size_t)ty * tiles_x);
	const int x = tx * 32 + cidx;
	uint32_t v = 0;
	if (x < W)
		for (int b = 0; b < 32; ++b) {
			const int y = ty * 32 + b;
			if (y < H && edge[(size_t)y * W + x] == equals) v |= 1u << b;
		}
	return v;
}
// smallest t in [a, b] (a <= b, both inside the image) with the bit of (t, fixed) set, or -1.  TRANSPOSED = false:
// t runs along x in row `fixed` of the row-major tiles; true: t runs along y in column `fixed` of the transposed tiles.
template <bool TRANSPOSED>
DVP_HD int first_set_bit_in(const uint32_t* bits, int tiles_x, int fixed, int a, int b) {
	for (int w0 = a >> 5; w0 <= (b >> 5); ++w0) {
		const size_t word = TRANSPOSED ? (size_t)((w0 * tiles_x + (fixed >> 5)) * 32 + (fixed & 31))
		                               : (size_t)(((fixed >> 5) * tiles_x + w0) * 32 + (fixed & 31));
		uint32_t v = bits[word];
		const int lo = w0 * 32;
		if (a > lo) v &= 0xFFFFFFFFu << (a - lo);
		if (b < lo + 31) v &= 0xFFFFFFFFu >> (lo + 31 - b);
		if (v) return lo + __builtin_ctz(v);
	}
	return -1;
}

DVP_HD int dvp_ctz(uint32_t v) { return __builtin_ctz(v); }   // v != 0
DVP_HD uint32_t f32_bits(float x) { return __builtin_bit_cast(uint32_t, x); }

// ---- small geometry helpers (APD.cu:181-194, 331-422, 467-499, 750-768) ----------------------
DVP_HD int is_set(uint32_t v, unsigned n) { return (v >> n) & 1; }
DVP_HD void set_bit(uint32_t* v, unsigned n) { *v |= (1u << n); }
DVP_HD void unset_bit_ref(uint32_t* v, unsigned n) { *v &= (0xFFFFFFFEu << n); }   // APD.cu:186-189: clears bits 0..n

DVP_HD void normalize3(f4* v) {   // NormalizeVec3 with rsqrtf -> 1/sqrtf
	const float n2 = v->x * v->x + v->y * v->y + v->z * v->z;
	const float inv = 1.0f / sqrtf(n2);
	v->x *= inv; v->y *= inv; v->z *= inv;
}
DVP_HD void normalize2(f2* v) {
	const float n2 = v->x * v->x + v->y * v->y;
	const float inv = 1.0f / sqrtf(n2);
	v->x *= inv; v->y *= inv;
}
DVP_HD void get_3d_point(const DvpCamera& cam, int px, int py, float depth, float* X) {   // APD.cu:372-377
	X[0] = depth * (px - cam.K[2]) / cam.K[0];
	X[1] = depth * (py - cam.K[5]) / cam.K[4];
	X[2] = depth;
}
DVP_HD f4 view_direction(const DvpCamera& cam, int px, int py, float depth) {              // APD.cu:386-398
	float X[3];
	get_3d_point(cam, px, py, depth, X);
	const float norm = sqrtf(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
	return mk4(X[0] / norm, X[1] / norm, X[2] / norm, 0.0f);
}
DVP_HD float distance_to_origin(const DvpCamera& cam, int px, int py, float depth, const f4 n) {   // APD.cu:400-405
	float X[3];
	get_3d_point(cam, px, py, depth, X);
	return -(n.x * X[0] + n.y * X[1] + n.z * X[2]);
}
DVP_HD float depth_from_plane(const DvpCamera& cam, const f4 pl, int px, int py) {          // APD.cu:419-422
	return -pl.w * cam.K[0] / ((px - cam.K[2]) * pl.x + (cam.K[0] / cam.K[4]) * (py - cam.K[5]) * pl.y + cam.K[0] * pl.z);
}
DVP_HD f3 point_on_world(float x, float y, float depth, const DvpCamera& cam) {              // APD.cu:467-487
	f3 P, T;
	P.x = depth * (x - cam.K[2]) / cam.K[0];
	P.y = depth * (y - cam.K[5]) / cam.K[4];
	P.z = depth;
	T.x = cam.R[0] * P.x + cam.R[3] * P.y + cam.R[6] * P.z;
	T.y = cam.R[1] * P.x + cam.R[4] * P.y + cam.R[7] * P.z;
	T.z = cam.R[2] * P.x + cam.R[5] * P.y + cam.R[8] * P.z;
	P.x = T.x + cam.c[0];
	P.y = T.y + cam.c[1];
	P.z = T.z + cam.c[2];
	return P;
}
DVP_HD void project_on_camera(const f3 X, const DvpCamera& cam, f2* pt, float* depth) {      // APD.cu:489-499
	f3 t;
	t.x = cam.R[0] * X.x + cam.R[1] * X.y + cam.R[2] * X.z + cam.t[0];
	t.y = cam.R[3] * X.x + cam.R[4] * X.y + cam.R[5] * X.z + cam.t[1];
	t.z = cam.R[6] * X.x + cam.R[7] * X.y + cam.R[8] * X.z + cam.t[2];
	*depth = cam.K[6] * t.x + cam.K[7] * t.y + cam.K[8] * t.z;
	pt->x = (cam.K[0] * t.x + cam.K[1] * t.y + cam.K[2] * t.z) / *depth;
	pt->y = (cam.K[3] * t.x + cam.K[4] * t.y + cam.K[5] * t.z) / *depth;
}
DVP_HD f4 normal_cam_to_world(const DvpCamera& cam, const f4 pl) {   // TransformNormal, APD.cu:750-758
	return mk4(cam.R[0] * pl.x + cam.R[3] * pl.y + cam.R[6] * pl.z,
	           cam.R[1] * pl.x + cam.R[4] * pl.y + cam.R[7] * pl.z,
	           cam.R[2] * pl.x + cam.R[5] * pl.y + cam.R[8] * pl.z, pl.w);
}
DVP_HD f4 normal_world_to_cam(const DvpCamera& cam, const f4 pl) {   // TransformNormal2RefCam, APD.cu:760-768
	return mk4(cam.R[0] * pl.x + cam.R[1] * pl.y + cam.R[2] * pl.z,
	           cam.R[3] * pl.x + cam.R[4] * pl.y + cam.R[5] * pl.z,
	           cam.R[6] * pl.x + cam.R[7] * pl.y + cam.R[8] * pl.z, pl.w);
}

// camera-only part of ComputeHomography (APD.cu:681-707), once per source view
DVP_HD void compute_view_const(const DvpCamera& ref, const DvpCamera& src, ViewConst* vc) {
	float ref_C[3], src_C[3];
	for (int j = 0; j < 3; ++j) {
		ref_C[j] = -(ref.R[j] * ref.t[0] + ref.R[3 + j] * ref.t[1] + ref.R[6 + j] * ref.t[2]);
		src_C[j] = -(src.R[j] * src.t[0] + src.R[3 + j] * src.t[1] + src.R[6 + j] * src.t[2]);
	}
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			vc->Rrel[3 * i + j] = src.R[3 * i] * ref.R[3 * j] + src.R[3 * i + 1] * ref.R[3 * j + 1] + src.R[3 * i + 2] * ref.R[3 * j + 2];
	float C_rel[3];
	for (int i = 0; i < 3; ++i) C_rel[i] = ref_C[i] - src_C[i];
	for (int i = 0; i < 3; ++i)
		vc->trel[i] = src.R[3 * i] * C_rel[0] + src.R[3 * i + 1] * C_rel[1] + src.R[3 * i + 2] * C_rel[2];
	vc->sK0 = src.K[0]; vc->sK2 = src.K[2]; vc->sK4 = src.K[4]; vc->sK5 = src.K[5]; vc->sK8 = src.K[8];
	vc->rK2 = ref.K[2]; vc->rK5 = ref.K[5];
	vc->inv_k0 = 1.0f / ref.K[0];
	vc->inv_k4 = 1.0f / ref.K[4];
	vc->fw = (float)src.width;
	vc->fh = (float)src.height;
	vc->pad = 0.0f;
}

// The per-view record of view `v`, which is the same for every lane that is active at the call site
// (v is a loop counter): on the device it is fetched through the constant address space with a
// wave-uniform address, i.e. with scalar loads into SGPRs instead of ~23 vector loads per evaluation
// (the compiler cannot prove the buffer read-only and emits global_load for a plain `d.views[v]`).
DVP_HD ViewConst load_view(const Dev& d, int v);

DVP_HD ViewConst load_view(const Dev& d, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
	const int vu = __builtin_amdgcn_readfirstlane(v);
	typedef const __attribute__((address_space(4))) ViewConst* cptr;
	return *(cptr)(d.views + vu);
#else
	return d.views[v];
#endif
}

// pins a wave-uniform float in an SGPR (an opaque value: the compiler cannot re-load it from memory
// right before its use, as it does with rematerialisable constant-address-space loads)
// true when the predicate holds on every active lane of the wave (host emulation: one lane at a time)
DVP_HD bool wave_all(bool pred) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __all(pred ? 1 : 0) != 0;
#else
	return pred;
#endif
}
DVP_HD float uniform_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
#else
	return x;
#endif
}

// element `i` (wave-uniform) of a read-only int table through the constant address space: s_load
DVP_HD int uniform_load_i32(const int* p, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
	typedef const __attribute__((address_space(4))) int* cptr;
	return *((cptr)p + __builtin_amdgcn_readfirstlane(i));
#else
	return p[i];
#endif
}

DVP_HD int uniform_i(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_readfirstlane(x);
#else
	return x;
#endif
}

// A camera record (112 B) by value through the same path; `v` must be wave-uniform (0 or a loop counter).
DVP_HD DvpCamera load_camera(const Dev& d, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
	const int vu = __builtin_amdgcn_readfirstlane(v);
	typedef const __attribute__((address_space(4))) DvpCamera* cptr;
	return *(cptr)(d.cameras + vu);
#else
	return d.cameras[v];
#endif
}

// plane-dependent part of ComputeHomography (APD.cu:709-738)
DVP_HD void homography(const ViewConst& vc, const f4 pl, float* H) {
	// a/b -> a * (1/b) with one correctly rounded reciprocal per divisor (numerics contract:
	// nvcc --use_fast_math lowers these divisions to a*rcp(b)); 1/K[0], 1/K[4] come with the record
	float Hh[9], tmp[9];
	const float inv_w = 1.0f / pl.w;
	const float inv_k0 = vc.inv_k0;
	const float inv_k4 = vc.inv_k4;
	for (int i = 0; i < 3; ++i) {
		Hh[3 * i + 0] = vc.Rrel[3 * i + 0] - vc.trel[i] * pl.x * inv_w;
		Hh[3 * i + 1] = vc.Rrel[3 * i + 1] - vc.trel[i] * pl.y * inv_w;
		Hh[3 * i + 2] = vc.Rrel[3 * i + 2] - vc.trel[i] * pl.z * inv_w;
	}
	for (int i = 0; i < 3; ++i) {
		tmp[3 * i + 0] = Hh[3 * i + 0] * inv_k0;
		tmp[3 * i + 1] = Hh[3 * i + 1] * inv_k4;
		tmp[3 * i + 2] = -Hh[3 * i + 0] * vc.rK2 * inv_k0 - Hh[3 * i + 1] * vc.rK5 * inv_k4 + Hh[3 * i + 2];
	}
	H[0] = vc.sK0 * tmp[0] + vc.sK2 * tmp[6];
	H[1] = vc.sK0 * tmp[1] + vc.sK2 * tmp[7];
	H[2] = vc.sK0 * tmp[2] + vc.sK2 * tmp[8];
	H[3] = vc.sK4 * tmp[3] + vc.sK5 * tmp[6];
	H[4] = vc.sK4 * tmp[4] + vc.sK5 * tmp[7];
	H[5] = vc.sK4 * tmp[5] + vc.sK5 * tmp[8];
	H[6] = vc.sK8 * tmp[6];
	H[7] = vc.sK8 * tmp[7];
	H[8] = vc.sK8 * tmp[8];
}
DVP_HD f2 apply_homography(const float* H, int px, int py) {   // ComputeCorrespondingPoint, APD.cu:741-748
	const float x = H[0] * px + H[1] * py + H[2];
	const float y = H[3] * px + H[4] * py + H[5];
	const float z = H[6] * px + H[7] * py + H[8];
	const float inv_z = 1.0f / z;
	return mk2(x * inv_z, y * inv_z);
}

}  // namespace dvp
#endif
