// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// NCC / geometric cost functions: APD.cu:835-1256.
#include "ora_core.h"

namespace ora {

// Weighted moments of one square patch around `c` (the tap loops of APD.cu:1059-1089 and 905-935).
//   numerics 0 (contract): the source's STRUCTURE — six partial sums of six taps each, added to the totals in order —
//     applied to the TRANSPOSED walk: y offset outer, x offset inner (the source walks x outer, y inner).  The walk
//     is transposed because a row of taps lies in ONE row-pair line of the image planes: walking columns re-fetches
//     every line six times for lanes with unrelated hypotheses (measured on MI355X, r03: strong update 592 -> 686 ms per
//     cfg3 pass; NCC costs move by 3e-6 median, contract-vs-literal end to end 2.7 % -> 2.6 % of pixels > 1e-3, i.e.
//     nothing).  The projective divide of a row is taken six taps at a time (batch_rcp); the three source-side sums use
//     one fused multiply-add per tap; the sampler takes pixel coordinates.
//   numerics 1 (literal): the reference's own order — x offset outer, y offset inner, the six taps of
//     one x offset summed into per-outer-index partials ("sum_*_row") that are then added to the
//     totals, one division per tap, tex2D(x + 0.5f, y + 0.5f) — every operator rounded once.
struct PatchSums { float ref, ref_ref, src, src_src, ref_src, w; };
template <class WeightFn>
static PatchSums patch_sums(const Ctx& h, const float* ref_image, const float* src_image, const float* H, const int2 c,
                            int radius, int increment, WeightFn weight_of) {
	const int W = h.width, Hh = h.height;
	PatchSums s = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
	if (h.numerics == 1) {
		for (int i = -radius; i <= radius; i += increment) {
			PatchSums r = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };   // sum_*_row of APD.cu:1060-1065 / 906-911
			for (int j = -radius; j <= radius; j += increment) {
				const int2 ref_pt = make_int2(c.x + i, c.y + j);
				const float ref_pix = tex_texel(ref_image, W, Hh, ref_pt.x, ref_pt.y);
				const float2 src_pt = ComputeCorrespondingPoint(H, ref_pt);
				const float src_pix = tex_linear_literal(src_image, W, Hh, src_pt.x + 0.5f, src_pt.y + 0.5f, h.sampler);
				const float weight = weight_of((float)i, (float)j, ref_pix);
				r.ref += weight * ref_pix;
				r.ref_ref += weight * ref_pix * ref_pix;
				r.src += weight * src_pix;
				r.src_src += weight * src_pix * src_pix;
				r.ref_src += weight * ref_pix * src_pix;
				r.w += weight;
			}
			s.ref += r.ref; s.ref_ref += r.ref_ref; s.src += r.src; s.src_src += r.src_src; s.ref_src += r.ref_src; s.w += r.w;   // APD.cu:1083-1088 / 920-925
		}
		return s;
	}
	for (int j = -radius; j <= radius; j += increment) {
		PatchSums r = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
		for (int i0 = -radius; i0 <= radius; i0 += 6 * increment) {
			float X[6], Y[6], Z[6], IZ[6];
			int off[6];
			int n = 0;
			for (int i = i0; i <= radius && n < 6; i += increment, ++n) {
				const int2 q = make_int2(c.x + i, c.y + j);
				X[n] = H[0] * q.x + H[1] * q.y + H[2];   // ComputeCorrespondingPoint, APD.cu:744-746
				Y[n] = H[3] * q.x + H[4] * q.y + H[5];
				Z[n] = H[6] * q.x + H[7] * q.y + H[8];
				off[n] = i;
			}
			batch_rcp(Z, n, IZ);
			for (int k = 0; k < n; ++k) {
				const float ref_pix = tex_texel(ref_image, W, Hh, c.x + off[k], c.y + j);
				const float src_pix = tex_linear(src_image, W, Hh, X[k] * IZ[k], Y[k] * IZ[k], h.sampler);
				const float weight = weight_of((float)off[k], (float)j, ref_pix);
				const float wa = weight * ref_pix, wb = weight * src_pix;
				r.ref += wa;
				r.ref_ref += wa * ref_pix;
				r.src += wb;
				r.src_src = fmaf(wb, src_pix, r.src_src);
				r.ref_src = fmaf(wa, src_pix, r.ref_src);
				r.w += weight;
			}
		}
		s.ref += r.ref; s.ref_ref += r.ref_ref; s.src += r.src; s.src_src += r.src_src; s.ref_src += r.ref_src; s.w += r.w;
	}
	return s;
}

// APD.cu:1023-1113
float ComputeBilateralNCCOld(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h) {
	const float* ref_image = h.images[0].data();
	const Camera& ref_camera = h.cameras[0];
	const float* src_image = h.images[src_idx].data();
	const Camera& src_camera = h.cameras[src_idx];
	const int W = h.width, Hh = h.height;
	const float cost_max = 2.0f;
	if (h.count_evals) {
#pragma omp atomic
		h.ncc_evals++;
	}

	float H[9];
	ComputeHomography(ref_camera, src_camera, plane_hypothesis, H);
	float2 pt = ComputeCorrespondingPoint(H, p);
	if (pt.x >= src_camera.width || pt.x < 0.0f || pt.y >= src_camera.height || pt.y < 0.0f) {
		return cost_max;
	}
	int radius = h.params.strong_radius;
	int increment = h.params.strong_increment;
	if (h.params.use_radius) {
		radius = h.radius[p.x + p.y * W];
		increment = ORA_MAX(2, (int)(2.0 * radius / 5.0));
	}

	float cost = 0.0f;
	{
		float sum_ref = 0.0f, sum_ref_ref = 0.0f, sum_src = 0.0f, sum_src_src = 0.0f, sum_ref_src = 0.0f;
		float bilateral_weight_sum = 0.0f;
		const float ref_center_pix = tex_texel(ref_image, W, Hh, p.x, p.y);

		const float sigma_spatial = h.params.sigma_spatial, sigma_color = h.params.sigma_color;
		const PatchSums ps = patch_sums(h, ref_image, src_image, H, p, radius, increment,
			[&](float xd, float yd, float pix) { return ComputeBilateralWeight(xd, yd, pix, ref_center_pix, sigma_spatial, sigma_color); });
		sum_ref = ps.ref; sum_ref_ref = ps.ref_ref; sum_src = ps.src; sum_src_src = ps.src_src; sum_ref_src = ps.ref_src;
		bilateral_weight_sum = ps.w;
		const float inv_bilateral_weight_sum = 1.0f / bilateral_weight_sum;
		sum_ref *= inv_bilateral_weight_sum;
		sum_ref_ref *= inv_bilateral_weight_sum;
		sum_src *= inv_bilateral_weight_sum;
		sum_src_src *= inv_bilateral_weight_sum;
		sum_ref_src *= inv_bilateral_weight_sum;
		const float var_ref = sum_ref_ref - sum_ref * sum_ref;
		const float var_src = sum_src_src - sum_src * sum_src;
		const float kMinVar = 1e-5f;
		if (var_ref < kMinVar || var_src < kMinVar) {
			cost = cost_max;
		} else {
			const float covar_src_ref = sum_ref_src - sum_ref * sum_src;
			const float var_ref_src = sqrtf(var_ref * var_src);
			cost = fmaxf(0.0f, fminf(cost_max, 1.0f - covar_src_ref / var_ref_src));
		}
	}
	return cost;
}

// APD.cu:835-1021.  Only valid for WEAK pixels (the reference printf("error") otherwise, :1017).
float ComputeBilateralNCCNew(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h) {
	const float* ref_image = h.images[0].data();
	const Camera& ref_camera = h.cameras[0];
	const float* src_image = h.images[src_idx].data();
	const Camera& src_camera = h.cameras[src_idx];
	const PatchMatchParams& params = h.params;
	const int width = h.width, height = h.height;
	const int S = h.num_images - 1;
	const float cost_max = 2.0f;
	if (h.count_evals) {
#pragma omp atomic
		h.ncc_evals++;
	}

	float H[9];
	ComputeHomography(ref_camera, src_camera, plane_hypothesis, H);
	float2 pt = ComputeCorrespondingPoint(H, p);
	if (pt.x >= src_camera.width || pt.x < 0.0f || pt.y >= src_camera.height || pt.y < 0.0f) {
		return cost_max;
	}
	float cost = 0.0f;
	const float ref_center_pix = tex_texel(ref_image, width, height, p.x, p.y);
	float center_cost = 0.0f, strong_cost = 0.0f;
	int strong_count = 0;
	for (int k = 0; k < NEIGHBOUR_NUM; ++k) {
		const short2 neighbour_pt = GetNeighbourPoint(p, k, h);
		if (neighbour_pt.x == -1 || neighbour_pt.y == -1) continue;
		float2 neighbour_src_pt = ComputeCorrespondingPoint(H, make_int2(neighbour_pt.x, neighbour_pt.y));
		if (neighbour_src_pt.x < 0 || neighbour_src_pt.y < 0 || neighbour_src_pt.x >= width || neighbour_src_pt.y >= height) {
			if (k != 0) {
				uint32_t view_info = h.selected_views[neighbour_pt.x + neighbour_pt.y * width];
				if (isSet(view_info, src_idx - 1)) {
					strong_cost += cost_max;
					strong_count++;
				}
				continue;
			} else {
				return cost_max;
			}
		}
		float sum_ref = 0.0f, sum_ref_ref = 0.0f, sum_src = 0.0f, sum_src_src = 0.0f, sum_ref_src = 0.0f;
		float bilateral_weight_sum = 0.0f;
		int radius = (k == 0 ? params.strong_radius : params.weak_radius);
		int increment = (k == 0 ? params.strong_increment : params.weak_increment);
		if (params.use_radius && k == 0) {
			radius = h.radius[p.x + p.y * width];
			increment = ORA_MAX(2, (int)(2.0 * radius / 5.0));
		}
		if (k == 0) {
			const PatchSums ps = patch_sums(h, ref_image, src_image, H, make_int2(neighbour_pt.x, neighbour_pt.y), radius, increment,
				[&](float xd, float yd, float pix) { return ComputeBilateralWeight_YZL(xd, yd, pix, ref_center_pix, params.sigma_spatial, params.sigma_color); });
			sum_ref = ps.ref; sum_ref_ref = ps.ref_ref; sum_src = ps.src; sum_src_src = ps.src_src; sum_ref_src = ps.ref_src;
			bilateral_weight_sum = ps.w;
		} else {
			if (isSet(h.selected_views[neighbour_pt.x + neighbour_pt.y * width], src_idx - 1) == 1) {
				int nei_center = neighbour_pt.x + neighbour_pt.y * width;
				for (int kk = 0; kk < 9; kk++) {
					int i = 0, j = 0;
					if (kk != 8) {
						// reference index: nei_center*8*NUM_IMAGES + (src_idx-1)*8 + kk with NUM_IMAGES=4
						// (APD.cu:940, aliases for S>4); here the per-pixel stride is S.
						const short2 c = h.candidate[((size_t)nei_center * S + (src_idx - 1)) * LAB_BOUNDARY_NUM + kk];
						i = c.x;
						j = c.y;
					}
					if (i == 0 && j == 0) {
						if (kk == 0) { i = -5; j = -5; }
						else if (kk == 1) { i = -5; j = 0; }
						else if (kk == 2) { i = -5; j = 5; }
						else if (kk == 3) { i = 0; j = -5; }
						else if (kk == 4) { i = 0; j = 5; }
						else if (kk == 5) { i = 5; j = -5; }
						else if (kk == 6) { i = 5; j = 0; }
						else if (kk == 7) { i = 5; j = 5; }
					}
					const int2 ref_pt = make_int2(neighbour_pt.x + i, neighbour_pt.y + j);
					const float ref_pix = tex_texel(ref_image, width, height, ref_pt.x, ref_pt.y);
					float2 src_pt = ComputeCorrespondingPoint(H, ref_pt);
					float weight = ComputeBilateralWeight_YZL((float)i, (float)j, ref_pix, ref_center_pix, params.sigma_spatial, params.sigma_color);
					if (h.numerics == 1) {
						const float src_pix = tex_linear_literal(src_image, width, height, src_pt.x + 0.5f, src_pt.y + 0.5f, h.sampler);
						sum_ref += weight * ref_pix;
						sum_ref_ref += weight * ref_pix * ref_pix;
						sum_src += weight * src_pix;
						sum_src_src += weight * src_pix * src_pix;
						sum_ref_src += weight * ref_pix * src_pix;
						bilateral_weight_sum += weight;
					} else {   // contract: fused multiply-adds on the source-side sums (one division per tap here)
						const float src_pix = tex_linear(src_image, width, height, src_pt.x, src_pt.y, h.sampler);
						const float wa = weight * ref_pix, wb = weight * src_pix;
						sum_ref += wa;
						sum_ref_ref += wa * ref_pix;
						sum_src += wb;
						sum_src_src = fmaf(wb, src_pix, sum_src_src);
						sum_ref_src = fmaf(wa, src_pix, sum_ref_src);
						bilateral_weight_sum += weight;
					}
				}
			}
		}
		// When the anchor is not visible in this view all sums are 0: 1/0 = inf, 0*inf = NaN,
		// the kMinVar test is false for NaN and CUDA's fminf/fmaxf drop the NaN: cost == 2.
		const float inv_bilateral_weight_sum = 1.0f / bilateral_weight_sum;
		sum_ref *= inv_bilateral_weight_sum;
		sum_ref_ref *= inv_bilateral_weight_sum;
		sum_src *= inv_bilateral_weight_sum;
		sum_src_src *= inv_bilateral_weight_sum;
		sum_ref_src *= inv_bilateral_weight_sum;
		const float var_ref = sum_ref_ref - sum_ref * sum_ref;
		const float var_src = sum_src_src - sum_src * sum_src;
		const float kMinVar = 1e-5f;
		float temp_cost = 0.0f;
		if (var_ref < kMinVar || var_src < kMinVar) {
			temp_cost = cost_max;
		} else {
			const float covar_src_ref = sum_ref_src - sum_ref * sum_src;
			const float var_ref_src = sqrtf(var_ref * var_src);
			temp_cost = fmaxf(0.0f, fminf(cost_max, 1.0f - covar_src_ref / var_ref_src));
		}
		if (k == 0) {
			center_cost = temp_cost;
		} else {
			strong_cost += temp_cost;
			strong_count++;
		}
	}
	if (strong_count == 0) {
		cost = center_cost;
	} else {
		strong_cost /= strong_count;
		strong_cost = ORA_MIN(strong_cost, cost_max);
		cost = (float)(0.25 * center_cost + 0.75 * strong_cost);   // double arithmetic, APD.cu:1013
	}
	return cost;
}

// APD.cu:1218-1256
float ComputeGeomConsistencyCost(const int2 p, const int src_idx, const float4 plane_hypothesis, Ctx& h) {
	const Camera& ref_camera = h.cameras[0];
	const Camera& src_camera = h.cameras[src_idx];
	const float* depth_image = h.depths[src_idx].data();
	const float max_cost = 3.0f;
	float center_cost = 0.0f;
	{
		float depth = ComputeDepthfromPlaneHypothesis(ref_camera, plane_hypothesis, p);
		float3 forward_point = Get3DPointonWorld_cu((float)p.x, (float)p.y, depth, ref_camera);
		float2 src_pt;
		float src_d;
		ProjectonCamera_cu(forward_point, src_camera, src_pt, src_d);
		// tex2D(depth, (int)x + 0.5f, (int)y + 0.5f): exact texel, clamp.  (int) of a NaN/huge
		// value is undefined in C++; defined here as clamp-to-range first (NaN -> 0).
		const float cx = fminf(fmaxf(src_pt.x, -1.0f), (float)h.width);
		const float cy = fminf(fmaxf(src_pt.y, -1.0f), (float)h.height);
		const float src_depth = tex_texel(depth_image, h.width, h.height, (int)cx, (int)cy);
		if (src_depth == 0.0f) return max_cost;
		float3 src_3D_pt = Get3DPointonWorld_cu(src_pt.x, src_pt.y, src_depth, src_camera);
		float2 backward_point;
		float ref_d;
		ProjectonCamera_cu(src_3D_pt, ref_camera, backward_point, ref_d);
		const float diff_col = p.x - backward_point.x;
		const float diff_row = p.y - backward_point.y;
		center_cost = sqrtf(diff_col * diff_col + diff_row * diff_row);
	}
	return fminf(max_cost, center_cost);
}

}  // namespace ora
