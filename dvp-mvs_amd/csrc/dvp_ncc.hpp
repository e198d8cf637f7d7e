// dvp_ncc.hpp — bilateral-weighted NCC cost (the roofline kernel core) with the
// hypothesis-independent half hoisted out.
//
// ComputeBilateralNCCOld (APD.cu:1023-1113) evaluates, per (pixel, view, plane): 36 reference
// texels, 36 bilateral weights (exp + sqrt each), the reference moments, and 36 bilinear source
// samples.  Everything that touches only the reference image depends on the pixel alone, yet the
// reference recomputes it for each of the ~20·S evaluations per pixel and iteration.  Here a
// PatchCtx is built once per pixel per kernel (weights w[t], w[t]*ref[t], the three reference
// moments in the reference's row-then-total accumulation order) and each evaluation only does the
// homography, the 36 gathers and three accumulations.  Results are bit-identical to the direct
// form because every floating-point operation that remains is the same operation on the same
// operands in the same order.
#ifndef DVP_NCC_HPP_
#define DVP_NCC_HPP_

#include "dvp_dev.hpp"

namespace dvp {

constexpr int kTaps = 6;   // taps per axis on the fast path (radius 5k, increment 2k)

// Per-lane table of the 36 (w, w*ref) pairs.  On the GPU it lives in LDS, laid out [tap][lane]
// (8-byte entries, lane-contiguous: conflict-free ds_read_b64), 72 KiB per 256-thread workgroup,
// which frees 72 VGPRs per lane and lets two waves share a SIMD without spilling.  The host
// emulation points it at a per-thread array (stride 1).
struct PatchTab {
	f2* p;
	int stride;
	DVP_HD f2 get(int t) const { return p[t * stride]; }
	DVP_HD void set(int t, f2 v) const { p[t * stride] = v; }
};

struct PatchCtx {
	PatchTab tab;              // (bilateral weight w, w * ref_pix) of tap (tx, ty) at index ty*6+tx (row-major: ty = y offset index)
	float sum_ref, sum_ref_ref, wsum;   // un-normalised reference sums (row-then-total order)
	int radius, inc;
	int fast;                  // 1: exactly 6 taps per axis (register path); 0: generic loops
};

// weight of ComputeBilateralWeight (APD.cu:776-781) / ComputeBilateralWeight_YZL (APD.cu:783-788)
DVP_HD float bilateral_weight(float xd, float yd, float pix, float cpix, float sig_s, float sig_c, int colour_only) {
	const float color_dist = fabsf(pix - cpix);
	if (colour_only) return dvp_expf(-color_dist / (2.0f * sig_c * sig_c));
	const float spatial_dist = sqrtf(xd * xd + yd * yd);
	return dvp_expf(-spatial_dist / (2.0f * sig_s * sig_s) - color_dist / (2.0f * sig_c * sig_c));
}

// radius / increment selection of APD.cu:1042-1047 (and 896-902 for the centre patch of NCCNew)
DVP_HD void patch_geometry(const Dev& d, int center, int* radius, int* inc) {
	int r = d.params.strong_radius, s = d.params.strong_increment;
	if (d.params.use_radius) {
		r = d.radius[center];
		s = DVP_MAX(2, (int)(2.0 * r / 5.0));
	}
	*radius = r;
	*inc = s;
}

DVP_HD void build_patch_ctx(const Dev& d, int px, int py, int radius, int inc, int colour_only, PatchTab tab, PatchCtx* c) {
	c->tab = tab;
	const float* ref = d.images;
	const int W = d.width, H = d.height, P = d.pitch;
	c->radius = radius;
	c->inc = inc;
	c->fast = (inc > 0 && (2 * radius) / inc + 1 == kTaps) ? 1 : 0;
	if (!c->fast) return;
	const float cpix = img_texel(ref, d.org, P, W, H, px, py);
	const float sig_s = d.params.sigma_spatial, sig_c = d.params.sigma_color;
	// the 36 reference texels first (independent loads, all in flight together), then the weights
	float av[kTaps * kTaps];
#pragma unroll
	for (int ty = 0; ty < kTaps; ++ty)
#pragma unroll
		for (int tx = 0; tx < kTaps; ++tx)
			av[ty * kTaps + tx] = img_texel(ref, d.org, P, W, H, px - radius + tx * inc, py - radius + ty * inc);
	sched_fence();
	// The spatial half of the weight, -sqrt(i^2 + j^2) / (2 sigma_s^2), depends on the tap offsets only.  With 2 radius = 5 inc
	// (the default 5 / 2 and every adaptive radius: multiples of 5 with inc = 2 radius / 5) the six offsets per axis are
	// -r .. r symmetric, and i^2 + j^2 = j^2 + i^2 in binary32 too: 6 distinct values instead of 36 square roots and divisions
	// per table — the same bits, a quarter of a table build's instructions (round 6).
	float sterm[3][3];
	const bool sym = wave_all(!colour_only && 2 * radius == 5 * inc);
	if (sym) {
#pragma unroll
		for (int a = 0; a < 3; ++a)
#pragma unroll
			for (int b = a; b < 3; ++b) {
				const float xd = (float)(-radius + a * inc), yd = (float)(-radius + b * inc);
				const float spatial_dist = sqrtf(xd * xd + yd * yd);
				sterm[a][b] = sterm[b][a] = -spatial_dist / (2.0f * sig_s * sig_s);
			}
	}
	float sr = 0.0f, srr = 0.0f, ws = 0.0f;
#pragma unroll
	for (int ty = 0; ty < kTaps; ++ty) {          // rows outer, columns inner (tap order)
		const int j = -radius + ty * inc;
		float sr_row = 0.0f, srr_row = 0.0f, ws_row = 0.0f;
#pragma unroll
		for (int tx = 0; tx < kTaps; ++tx) {
			const int i = -radius + tx * inc;
			const float a = av[ty * kTaps + tx];
			const float w = sym ? dvp_expf(sterm[tx < 3 ? tx : 5 - tx][ty < 3 ? ty : 5 - ty] - fabsf(a - cpix) / (2.0f * sig_c * sig_c))
			                    : bilateral_weight((float)i, (float)j, a, cpix, sig_s, sig_c, colour_only);
			const float wa = w * a;
			tab.set(ty * kTaps + tx, mk2(w, wa));
			sr_row += wa;
			srr_row += wa * a;
			ws_row += w;
		}
		sr += sr_row;
		srr += srr_row;
		ws += ws_row;
	}
	c->sum_ref = sr;
	c->sum_ref_ref = srr;
	c->wsum = ws;
}

// final NCC formula of APD.cu:1091-1109 from the six un-normalised sums
DVP_HD float ncc_from_sums(float sum_ref, float sum_ref_ref, float sum_src, float sum_src_src, float sum_ref_src, float wsum) {
	const float inv = 1.0f / wsum;
	sum_ref *= inv;
	sum_ref_ref *= inv;
	sum_src *= inv;
	sum_src_src *= inv;
	sum_ref_src *= inv;
	const float var_ref = sum_ref_ref - sum_ref * sum_ref;
	const float var_src = sum_src_src - sum_src * sum_src;
	const float kMinVar = 1e-5f;
	if (var_ref < kMinVar || var_src < kMinVar) return 2.0f;
	const float covar = sum_ref_src - sum_ref * sum_src;
	const float denom = sqrtf(var_ref * var_src);
	return fmaxf(0.0f, fminf(2.0f, 1.0f - covar / denom));
}

// Generic (non-hoisted) patch loop for tap counts other than 6 per axis; restatement of
// APD.cu:1059-1089 with the weight variant selected by `colour_only` (taps visited row by row).
DVP_HD float ncc_patch_generic(const Dev& d, const float* H, const float* src, int px, int py, int radius, int inc, int colour_only) {
	const float* ref = d.images;
	const int W = d.width, Hh = d.height, P = d.pitch;
	const float cpix = img_texel(ref, d.org, P, W, Hh, px, py);
	float s_r = 0.0f, s_rr = 0.0f, s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f, s_w = 0.0f;
	if (inc <= 0) inc = 1;
	for (int j = -radius; j <= radius; j += inc) {        // rows outer, columns inner
		float r_r = 0.0f, r_rr = 0.0f, r_s = 0.0f, r_ss = 0.0f, r_rs = 0.0f, r_w = 0.0f;
		for (int i0 = -radius; i0 <= radius; i0 += 6 * inc) {   // projective divide: six taps at a time
			float X[6], Y[6], Z[6], IZ[6];
			int n = 0;
			for (int i = i0; i <= radius && n < 6; i += inc, ++n) {
				const int qx = px + i, qy = py + j;
				X[n] = H[0] * qx + H[1] * qy + H[2];
				Y[n] = H[3] * qx + H[4] * qy + H[5];
				Z[n] = H[6] * qx + H[7] * qy + H[8];
			}
			batch_rcp(Z, n, IZ);
			for (int k = 0; k < n; ++k) {
				const int i = i0 + k * inc;
				const float a = img_texel(ref, d.org, P, W, Hh, px + i, py + j);
				const float b = tex_linear(src, P, W, Hh, X[k] * IZ[k], Y[k] * IZ[k], d.sampler);
				const float w = bilateral_weight((float)i, (float)j, a, cpix, d.params.sigma_spatial, d.params.sigma_color, colour_only);
				const float wa = w * a, wb = w * b;
				r_r += wa;
				r_rr += wa * a;
				r_s += wb;
				r_ss = fmaf(wb, b, r_ss);
				r_rs = fmaf(wa, b, r_rs);
				r_w += w;
			}
		}
		s_r += r_r; s_rr += r_rr; s_s += r_s; s_ss += r_ss; s_rs += r_rs; s_w += r_w;
	}
	return ncc_from_sums(s_r, s_rr, s_s, s_ss, s_rs, s_w);
}

// 36-tap patch with the hoisted context.  H is the pixel->source homography.  The sampler mode is
// a template parameter so that the unrolled tap body is one branch-free basic block.
// CLAMP = false: every tap's source coordinate is known to lie in [-1, W] x [-1, H] (patch_stays_inside) — the sampler's
// clamp is then the identity and is left out; results are the same bits.
template <int SMP, bool CLAMP = true>
DVP_HD float ncc_patch_fast(const Dev& d, const PatchCtx& c, const float* H, const float* src, int px, int py, unsigned plane_off = 0) {
	const int W = d.width, Hh = d.height, P = d.pitch;
	// H[k]*x and H[k]*y products for the 6 distinct tap columns / rows: same products the
	// reference forms per tap (H[0]*p.x + H[1]*p.y + H[2], APD.cu:744-746), formed once.
	float hx0[kTaps], hx3[kTaps], hx6[kTaps];
#pragma unroll
	for (int t = 0; t < kTaps; ++t) {
		const float fx = (float)(px - c.radius + t * c.inc);
		hx0[t] = H[0] * fx; hx3[t] = H[3] * fx; hx6[t] = H[6] * fx;
	}
	// Software pipeline over the six ROWS of the patch (6 taps each), three buffers deep:
	//   coords(r): one batched reciprocal + 6 footprint addresses / weights
	//   issue(r) : 6 sixteen-byte gathers
	//   consume(r): 6 blends + the row sums, rows in order (row-then-total accumulation)
	// ordered  coords0 issue0 coords1 issue1 | coords2 | consume0 issue2 | coords3 | consume1 issue3 | ... :
	// the gathers of two rows (12) fly while the addresses of the row after them are computed, and
	// only the last row's latency stays exposed.  (Round 1 ran the same pipeline over row PAIRS, two
	// buffers deep: 256 VGPRs and 3 % slower; more rows in flight spill.)  Taps of one row land on
	// the same source row pair (near-upright homographies): a lane with a hypothesis unrelated to
	// its neighbours' touches 1-2 cache lines per row.
	unsigned off[3][kTaps];
	TapW<SMP> tw[3][kTaps];
	float q[3][kTaps][4];
	float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
#define DVP_COORDS(R, BUF)                                                                          \
	{                                                                                               \
		const float fy = (float)(py - c.radius + (R) * c.inc);                                      \
		const float hy1 = H[1] * fy, hy4 = H[4] * fy, hy7 = H[7] * fy;                              \
		float X[kTaps], Y[kTaps], Z[kTaps], IZ[kTaps];                                              \
		_Pragma("unroll") for (int tx = 0; tx < kTaps; ++tx) {                                      \
			X[tx] = hx0[tx] + hy1 + H[2];                                                           \
			Y[tx] = hx3[tx] + hy4 + H[5];                                                           \
			Z[tx] = hx6[tx] + hy7 + H[8];                                                           \
		}                                                                                           \
		batch_rcp(Z, kTaps, IZ);                                                                    \
		_Pragma("unroll") for (int tx = 0; tx < kTaps; ++tx)                                        \
			tex_coord<SMP, CLAMP>(P, W, Hh, X[tx] * IZ[tx], Y[tx] * IZ[tx], &off[BUF][tx], &tw[BUF][tx], plane_off); \
	}
#define DVP_ISSUE(BUF)                                                                              \
	_Pragma("unroll") for (int k = 0; k < kTaps; ++k)                                               \
		load_quad(src, off[BUF][k], &q[BUF][k][0], &q[BUF][k][1], &q[BUF][k][2], &q[BUF][k][3]);
#define DVP_CONSUME(R, BUF)                                                                         \
	{                                                                                               \
		f2 tt[kTaps];   /* the row's 6 table entries: all LDS reads issued before the first use */  \
		_Pragma("unroll") for (int k = 0; k < kTaps; ++k) tt[k] = c.tab.get((R) * kTaps + k);       \
		sched_fence();                                                                              \
		float r_s = 0.0f, r_ss = 0.0f, r_rs = 0.0f;                                                 \
		_Pragma("unroll") for (int k = 0; k < kTaps; ++k) {                                         \
			float fa, fb;                                                                           \
			tap_weights(tw[BUF][k], &fa, &fb);                                                      \
			const float b = tex_lerp(fa, fb, q[BUF][k][0], q[BUF][k][1], q[BUF][k][2], q[BUF][k][3]); \
			const f2 t = tt[k];                                                                     \
			const float wsb = t.x * b;                                                              \
			r_s += wsb;                                                                             \
			r_ss = fmaf(wsb, b, r_ss);                                                              \
			r_rs = fmaf(t.y, b, r_rs);                                                              \
		}                                                                                           \
		s_s += r_s;                                                                                 \
		s_ss += r_ss;                                                                               \
		s_rs += r_rs;                                                                               \
	}
	DVP_COORDS(0, 0) DVP_ISSUE(0) sched_fence();
	DVP_COORDS(1, 1) DVP_ISSUE(1) sched_fence();
	DVP_COORDS(2, 2) sched_fence();
	DVP_CONSUME(0, 0) DVP_ISSUE(2) sched_fence();
	DVP_COORDS(3, 0) sched_fence();
	DVP_CONSUME(1, 1) DVP_ISSUE(0) sched_fence();
	DVP_COORDS(4, 1) sched_fence();
	DVP_CONSUME(2, 2) DVP_ISSUE(1) sched_fence();
	DVP_COORDS(5, 2) sched_fence();
	DVP_CONSUME(3, 0) DVP_ISSUE(2) sched_fence();
	DVP_CONSUME(4, 1) sched_fence();
	DVP_CONSUME(5, 2)
#undef DVP_COORDS
#undef DVP_ISSUE
#undef DVP_CONSUME
	return ncc_from_sums(c.sum_ref, c.sum_ref_ref, s_s, s_ss, s_rs, c.wsum);
}

// Does the whole patch (|dx|, |dy| <= r around the pixel) project at least one pixel inside the source image?  With
// q(p) = (X, Y) / z, q(p + D) - q(p) = ((dX, dY) - q(p) dz) / (z + dz) where (dX, dY, dz) = H D, so
// |q.x(p + D) - q.x(p)| <= (r (|H0| + |H1|) + |q.x| r (|H6| + |H7|)) / (|z| - r (|H6| + |H7|)) whenever the denominator is
// positive (and likewise in y).  The test asks for a full pixel of slack on every side, far more than the rounding of
// the bound and of the tap coordinates themselves; NaN / inf anywhere makes a comparison false, i.e. the clamped path.
// NOT ENABLED (build with -DDVP_CLAMP_FREE): the clamp is 2 of ~42 VALU instructions per tap, but a second inlined copy
// of the 36-tap body next to the first pushes the 256-VGPR kernels into spilling (strong update 7 -> 83 spilled VGPRs,
// DepthToWeak 0 -> 91): measured at cfg3 (r03) 31.2 Mpx/s/iter with this path against 38.9 without (strong update 712 ->
// 804 ms, DepthToWeak 633 -> 1032 ms per pass); the clamped copy as a real function call is worse still (1098 / 953 ms).
DVP_HD bool patch_stays_inside(const float* H, const f2 pt, int px, int py, float r, float fw, float fh) {
	const float z = H[6] * px + H[7] * py + H[8];
	const float ez = r * (fabsf(H[6]) + fabsf(H[7]));
	const float zl = fabsf(z) - ez;
	const float bx = r * (fabsf(H[0]) + fabsf(H[1])) + fabsf(pt.x) * ez;
	const float by = r * (fabsf(H[3]) + fabsf(H[4])) + fabsf(pt.y) * ez;
	return zl > 0.0f && (pt.x - 1.0f) * zl >= bx && (fw - 2.0f - pt.x) * zl >= bx && (pt.y - 1.0f) * zl >= by && (fh - 2.0f - pt.y) * zl >= by;
}

// ComputeBilateralNCCOld (APD.cu:1023-1113) for source view `v` (1-based image index).
template <int SMP>
DVP_HD float ncc_old(const Dev& d, const PatchCtx& c, int px, int py, int v, const f4 plane) {
	const ViewConst vc = load_view(d, v);
	const float fw = uniform_f(vc.fw), fh = uniform_f(vc.fh);
	float H[9];
	homography(vc, plane, H);
	const f2 pt = apply_homography(H, px, py);
	if (pt.x >= fw || pt.x < 0.0f || pt.y >= fh || pt.y < 0.0f) return 2.0f;
	const float* src = d.images + (size_t)uniform_i(v) * d.plane_stride * 2;   // wave-uniform base
#ifdef DVP_CLAMP_FREE
	if (c.fast && wave_all(patch_stays_inside(H, pt, px, py, (float)c.radius, (float)d.width, (float)d.height)))
		return ncc_patch_fast<SMP, false>(d, c, H, src, px, py);
#endif
	if (c.fast) return ncc_patch_fast<SMP>(d, c, H, src, px, py);
	return ncc_patch_generic(d, H, src, px, py, c.radius, c.inc, 0);
}

// The same for a LANE-VARYING view (dvp_strong_refine's per-lane walk): the view record comes through vector loads and the
// lane's image plane is a byte offset from the set's base (the caller checks that the whole set lies below 4 GiB).
template <int SMP>
DVP_HD float ncc_old_lane(const Dev& d, const PatchCtx& c, int px, int py, int v, const f4 plane) {
	const ViewConst vc = d.views[v];
	float H[9];
	homography(vc, plane, H);
	const f2 pt = apply_homography(H, px, py);
	if (pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f) return 2.0f;
	if (c.fast) return ncc_patch_fast<SMP>(d, c, H, d.images, px, py, (unsigned)((size_t)v * d.plane_stride * 8));
	return ncc_patch_generic(d, H, d.images + (size_t)v * d.plane_stride * 2, px, py, c.radius, c.inc, 0);
}

// ComputeGeomConsistencyCost (APD.cu:1218-1256) in two pieces: the forward point depends on the pixel and the
// plane only, so a caller that tests one plane against several views forms it once.
DVP_HD f3 geom_forward_point(const Dev& d, int px, int py, const f4 plane) {
	const DvpCamera rc = load_camera(d, 0);
	const float depth = depth_from_plane(rc, plane, px, py);
	return point_on_world((float)px, (float)py, depth, rc);
}
DVP_HD float geom_cost_of_point(const Dev& d, int px, int py, int v, const f3 fwd) {
	const DvpCamera rc = load_camera(d, 0);
	const DvpCamera sc = load_camera(d, v);
	const float* dimg = d.depths + (size_t)v * d.plane_stride;
	f2 sp;
	float sd;
	project_on_camera(fwd, sc, &sp, &sd);
	const float cx = fminf(fmaxf(sp.x, -1.0f), (float)d.width);
	const float cy = fminf(fmaxf(sp.y, -1.0f), (float)d.height);
	const float src_depth = tex_texel(dimg, d.org, d.pitch, d.width, d.height, (int)cx, (int)cy);
	if (src_depth == 0.0f) return 3.0f;
	const f3 back = point_on_world(sp.x, sp.y, src_depth, sc);
	f2 bp;
	float rd;
	project_on_camera(back, rc, &bp, &rd);
	const float dc = px - bp.x, dr = py - bp.y;
	return fminf(3.0f, sqrtf(dc * dc + dr * dr));
}
DVP_HD float geom_cost(const Dev& d, int px, int py, int v, const f4 plane) {
	return geom_cost_of_point(d, px, py, v, geom_forward_point(d, px, py, plane));
}

// ComputeGeomConsistencyCost (APD.cu:1218-1256) with both cameras given: a lane-varying view index (weak update), or records the caller keeps in LDS
// (dvp_sweep_eval: the constant-address-space loads of load_camera are re-issued as ~18 vector loads in three dependent groups per call)
DVP_HD float geom_cost_cams(const Dev& d, const DvpCamera& rc, const DvpCamera& sc, int v, int px, int py, const f4 plane) {
	const float* dimg = d.depths + (size_t)v * d.plane_stride;
	const float depth = depth_from_plane(rc, plane, px, py);
	const f3 fwd = point_on_world((float)px, (float)py, depth, rc);
	f2 sp;
	float sd;
	project_on_camera(fwd, sc, &sp, &sd);
	const float cx = fminf(fmaxf(sp.x, -1.0f), (float)d.width);
	const float cy = fminf(fmaxf(sp.y, -1.0f), (float)d.height);
	const float src_depth = tex_texel(dimg, d.org, d.pitch, d.width, d.height, (int)cx, (int)cy);
	if (src_depth == 0.0f) return 3.0f;
	const f3 back = point_on_world(sp.x, sp.y, src_depth, sc);
	f2 bp;
	float rd;
	project_on_camera(back, rc, &bp, &rd);
	const float dc = px - bp.x, dr = py - bp.y;
	return fminf(3.0f, sqrtf(dc * dc + dr * dr));
}

// geom_cost_cams in two halves, so that a caller can put an NCC evaluation between the depth-map fetch and its use (the fetch is
// a dependent load at the end of a chain of divisions: dvp_sweep_eval waited for it once per (slot, view)).  Same operations on
// the same operands in the same order as geom_cost_cams.
struct GeomFetch { f2 sp; float src_depth; };
DVP_HD GeomFetch geom_cost_fetch(const Dev& d, const DvpCamera& rc, const DvpCamera& sc, int v, int px, int py, const f4 plane) {
	const float* dimg = d.depths + (size_t)v * d.plane_stride;
	const float depth = depth_from_plane(rc, plane, px, py);
	const f3 fwd = point_on_world((float)px, (float)py, depth, rc);
	GeomFetch g;
	float sd;
	project_on_camera(fwd, sc, &g.sp, &sd);
	const float cx = fminf(fmaxf(g.sp.x, -1.0f), (float)d.width);
	const float cy = fminf(fmaxf(g.sp.y, -1.0f), (float)d.height);
	g.src_depth = tex_texel(dimg, d.org, d.pitch, d.width, d.height, (int)cx, (int)cy);
	return g;
}
DVP_HD float geom_cost_finish(const DvpCamera& rc, const DvpCamera& sc, int px, int py, const GeomFetch& g) {
	if (g.src_depth == 0.0f) return 3.0f;
	const f3 back = point_on_world(g.sp.x, g.sp.y, g.src_depth, sc);
	f2 bp;
	float rd;
	project_on_camera(back, rc, &bp, &rd);
	const float dc = px - bp.x, dr = py - bp.y;
	return fminf(3.0f, sqrtf(dc * dc + dr * dr));
}

}  // namespace dvp
#endif
