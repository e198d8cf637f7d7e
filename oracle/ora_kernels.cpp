// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// Per-pixel kernel bodies of the strong path + post-processing: APD.cu:501-669, 1115-1383,
// 2010-2141, 2462-2567, 2725-2737, 3127-3328, 3892-4139.
#include "ora_core.h"
#include "ora_kernels.h"

namespace ora {

// APD.cu:501-588.  `rng` is the SUB_NORMAL sub-stream of the calling site.
float4 GenerateRandomNormal_YZL(Ctx& h, const Camera& camera, const int2 p, Rng& rng, const float depth) {
	const int width = h.width, height = h.height;
	const int center = p.y * width + p.x;
	const PatchMatchParams& params = h.params;
	float4 view_direction[20];
	for (auto& v : view_direction) v = make_float4(0, 0, 0, 0);
	view_direction[0] = GetViewDirection(camera, p, depth);
	int index = 1;
	for (int src_idx = 1; src_idx < params.num_images; ++src_idx) {
		const Camera& ref_camera = h.cameras[0];
		const Camera& src_camera = h.cameras[src_idx];
		if (isSet(h.selected_views[center], src_idx - 1) == 1) {
			float3 forward_point = Get3DPointonWorld_cu((float)p.x, (float)p.y, depth, ref_camera);
			float2 src_pt;
			float src_d;
			ProjectonCamera_cu(forward_point, src_camera, src_pt, src_d);
			// make_int2((int)src_pt.x + 0.5f, ...) : int + 0.5f -> float -> int (APD.cu:525)
			const float sx = fminf(fmaxf(src_pt.x, -32768.0f), 32767.0f);
			const float sy = fminf(fmaxf(src_pt.y, -32768.0f), 32767.0f);
			int2 src_pt_int = make_int2((int)((float)(int)sx + 0.5f), (int)((float)(int)sy + 0.5f));
			float src_depth = 1.0f;   // APD.cu:526: uninitialised when outside the image; defined as 1
			if (params.geom_consistency) {
				if (src_pt_int.x >= 0 && src_pt_int.x < width && src_pt_int.y >= 0 && src_pt_int.y < height)
					src_depth = tex_texel(h.depths[src_idx].data(), width, height, (int)sx, (int)sy);
			}
			float4 direction = GetViewDirection(h.cameras[src_idx], src_pt_int, src_depth);
			float R_t[9], R_c[9], R_f[3];
			matTranspose3x3(h.cameras[src_idx].R, R_t);
			matMul3x3(h.cameras[0].R, R_t, R_c);
			float dir[3] = { direction.x, direction.y, direction.x };   // APD.cu:543 ({x,y,x}: kept)
			matMul3x1_ref(R_c, dir, R_f);
			float norm = sqrtf(R_f[0] * R_f[0] + R_f[1] * R_f[1] + R_f[2] * R_f[2]);
			float4 v = make_float4(R_f[0] / norm, R_f[1] / norm, R_f[2] / norm, 0);
			if (index < 20) view_direction[index++] = v;   // reference array has 20 slots (APD.cu:511)
		}
	}
	int times = 200;
	float4 normal = make_float4(0, 0, 0, 0);
	while (times > 0) {
		float q1 = 1.0f, q2 = 1.0f, s = 2.0f;
		while (s >= 1.0f) {
			q1 = 2.0f * rng.uniform() - 1.0f;
			q2 = 2.0f * rng.uniform() - 1.0f;
			s = q1 * q1 + q2 * q2;
		}
		const float sq = sqrtf(1.0f - s);
		normal.x = 2.0f * q1 * sq;
		normal.y = 2.0f * q2 * sq;
		normal.z = 1.0f - 2.0f * s;
		normal.w = 0;
		bool satisfy = true;
		for (int i = 0; i < index; i++) {
			float d = normal.x * view_direction[i].x + normal.y * view_direction[i].y + normal.z * view_direction[i].z;
			if (d > 0.0f) { satisfy = false; break; }
		}
		if (satisfy) break;
		else times--;
	}
	NormalizeVec3(&normal);
	return normal;
}

// APD.cu:617-661: the loop only consumes random numbers — both branches assign
// `normal_perturbed = normal` — so the result is the normalised input normal.  With the
// counter-based RNG the consumed draws are unobservable and are not generated.
static float4 GeneratePerturbedNormal(const float4 normal) {
	float4 n = normal;
	NormalizeVec3(&n);
	return n;
}

// APD.cu:663-669
static float4 GenerateRandomPlaneHypothesis_YZL(Ctx& h, const Camera& camera, const int2 p, int phase, int iter, const float depth_min, const float depth_max) {
	const uint32_t pix = (uint32_t)(p.y * h.width + p.x);
	Rng rd(h.seed, pix, rng_site(phase, iter, SUB_DEPTH_RAND));
	Rng rn(h.seed, pix, rng_site(phase, iter, SUB_NORMAL));
	float depth = rd.uniform() * (depth_max - depth_min) + depth_min;
	float4 pl = GenerateRandomNormal_YZL(h, camera, p, rn, depth);
	pl.w = GetDistance2Origin(camera, p, depth, pl);
	return pl;
}

// APD.cu:1115-1161
static float ComputeMultiViewInitialCostandSelectedViews(const int2 p, Ctx& h) {
	const PatchMatchParams& params = h.params;
	int center = p.x + p.y * h.width;
	float4 plane_hypothesis = h.planes[center];
	float cost_max = 2.0f;
	float cost_vector[32] = { 2.0f }, cost_vector_copy[32] = { 2.0f };
	int cost_count = 0, num_valid_views = 0;
	for (int i = 1; i < params.num_images; ++i) {
		float c = ComputeBilateralNCCOld(p, i, plane_hypothesis, h);
		cost_vector[i - 1] = c;
		cost_vector_copy[i - 1] = c;
		cost_count++;
		if (c < cost_max) num_valid_views++;
	}
	sort_small(cost_vector, cost_count);
	h.selected_views[center] = 0;
	int top_k = ORA_MIN(num_valid_views, params.top_k);
	if (top_k > 0) {
		float cost = 0.0f;
		for (int i = 0; i < top_k; ++i) cost += cost_vector[i];
		float cost_threshold = cost_vector[top_k - 1];
		for (int i = 0; i < params.num_images - 1; ++i)
			if (cost_vector_copy[i] <= cost_threshold) setBit(&(h.selected_views[center]), i);
		return cost / top_k;
	}
	return cost_max;
}

// APD.cu:1163-1194
static float ComputeMultiViewInitialCost(const int2 p, Ctx& h) {
	const PatchMatchParams& params = h.params;
	int center = p.x + p.y * h.width;
	float4 plane_hypothesis = h.planes[center];
	const float cost_max = 2.0f;
	int cost_count = 0;
	float cost = 0.0f;
	for (int i = 1; i < params.num_images; ++i) {
		if (isSet(h.selected_views[center], i - 1)) {
			float c = ComputeBilateralNCCOld(p, i, plane_hypothesis, h);
			if (c < cost_max) { cost_count++; cost += c; }
			else unSetBit(&(h.selected_views[center]), i - 1);
		}
	}
	if (cost_count == 0) return cost_max;
	return cost / cost_count;
}

void ComputeMultiViewCostVectorOld(const int2 p, float4 pl, float* cost_vector, Ctx& h) {   // APD.cu:1207-1216
	for (int i = 1; i < h.params.num_images; ++i) cost_vector[i - 1] = ComputeBilateralNCCOld(p, i, pl, h);
}
void ComputeMultiViewCostVectorNew(const int2 p, float4 pl, float* cost_vector, Ctx& h) {   // APD.cu:1196-1205
	for (int i = 1; i < h.params.num_images; ++i) cost_vector[i - 1] = ComputeBilateralNCCNew(p, i, pl, h);
}

// APD.cu:1273-1309
void RandomInitialization_px(Ctx& h, const int2 p) {
	const int center = p.y * h.width + p.x;
	const PatchMatchParams& params = h.params;
	if (params.state == FIRST_INIT) {
		// the prior's .w is a depth; if in range the plane is kept as is (world normal, depth as
		// offset) — APD.cu:1289-1295, reproduced.
		if (h.planes[center].w > params.depth_max || h.planes[center].w < params.depth_min)
			h.planes[center] = GenerateRandomPlaneHypothesis_YZL(h, h.cameras[0], p, PH_RANDOM_INIT, 0, params.depth_min, params.depth_max);
		h.costs[center] = ComputeMultiViewInitialCostandSelectedViews(p, h);
	} else {
		float4 pl = h.planes[center];
		pl = TransformNormal2RefCam(h.cameras[0], pl);
		float depth = pl.w;
		pl.w = GetDistance2Origin(h.cameras[0], p, depth, pl);
		h.planes[center] = pl;
		h.costs[center] = ComputeMultiViewInitialCost(p, h);
	}
}

// APD.cu:1311-1383 (strong) — hypotheses are built from the values at entry (:1359-1360).
static void PlaneHypothesisRefinementStrong(float4* plane_hypothesis, float* depth, float* cost, int iter,
	const uint8_t* view_weights, const float weight_norm, const int2 p, Ctx& h) {
	float depth_perturbation = 0.02f;
	const Camera* cameras = h.cameras;
	const PatchMatchParams& params = h.params;
	float depth_min = params.depth_min, depth_max = params.depth_max;
	const uint32_t pix = (uint32_t)(p.y * h.width + p.x);
	Rng rd(h.seed, pix, rng_site(PH_STRONG, iter, SUB_DEPTH_RAND));
	Rng rn(h.seed, pix, rng_site(PH_STRONG, iter, SUB_NORMAL));
	Rng rp(h.seed, pix, rng_site(PH_STRONG, iter, SUB_DEPTH_PERT));

	float depth_rand = rd.uniform() * (depth_max - depth_min) + depth_min;
	float4 plane_hypothesis_rand = GenerateRandomNormal_YZL(h, cameras[0], p, rn, *depth);
	float depth_perturbed = *depth;
	const float depth_min_perturbed = (1 - depth_perturbation) * depth_perturbed;
	const float depth_max_perturbed = (1 + depth_perturbation) * depth_perturbed;
	do {
		depth_perturbed = rp.uniform() * (depth_max_perturbed - depth_min_perturbed) + depth_min_perturbed;
	} while (depth_perturbed < depth_min && depth_perturbed > depth_max);   // never loops (:1340)
	float4 plane_hypothesis_perturbed = GeneratePerturbedNormal(*plane_hypothesis);
	float4 plane_hypothesis_perturbed_2 = GeneratePerturbedNormal(*plane_hypothesis);

	const int num_planes = 6;
	float depths[num_planes] = { depth_rand, *depth, depth_rand, *depth, *depth, depth_perturbed };
	float4 normals[num_planes] = { *plane_hypothesis, plane_hypothesis_rand, plane_hypothesis_rand, plane_hypothesis_perturbed, plane_hypothesis_perturbed_2, *plane_hypothesis };

	for (int i = 0; i < num_planes; ++i) {
		float cost_vector[32] = { 2.0f };
		float4 temp = normals[i];
		temp.w = GetDistance2Origin(cameras[0], p, depths[i], temp);
		ComputeMultiViewCostVectorOld(p, temp, cost_vector, h);
		float temp_cost = 0.0f;
		for (int j = 0; j < params.num_images - 1; ++j)
			if (view_weights[j] > 0) temp_cost += view_weights[j] * cost_vector[j];
		temp_cost /= weight_norm;
		float depth_before = ComputeDepthfromPlaneHypothesis(cameras[0], temp, p);
		if (depth_before >= depth_min && depth_before <= depth_max && temp_cost < *cost) {
			*depth = depth_before;
			*plane_hypothesis = temp;
			*cost = temp_cost;
		}
	}
}

// Multi-hypothesis joint view selection, shared verbatim by the strong (APD.cu:2462-2530) and
// weak (APD.cu:2781-2850) updates; only the prior loop differs and is done by the caller.
void JointViewSelection(Ctx& h, int center, int iter, int phase, float cost_array[8][32],
	const float* view_selection_priors, uint8_t* view_weights, uint32_t* temp_selected_views, float* weight_norm) {
	const int num_images = h.params.num_images;
	float sampling_probs[32] = { 0.0f };
	float cost_threshold = (float)(0.8 * dvp_expf((iter) * (iter) / (-90.0f)));   // 0.8 is a double literal (:2484)
	for (int i = 0; i < num_images - 1; i++) {
		float count = 0;
		int count_false = 0;
		float tmpw = 0;
		for (int j = 0; j < 8; j++) {
			if (cost_array[j][i] < cost_threshold) {
				tmpw += dvp_expf(cost_array[j][i] * cost_array[j][i] / (-0.18f));
				count++;
			}
			if (cost_array[j][i] > 1.2f) count_false++;
		}
		if (count > 2 && count_false < 3) sampling_probs[i] = tmpw / count;
		else if (count_false < 3) sampling_probs[i] = dvp_expf(cost_threshold * cost_threshold / (-0.32f));
		sampling_probs[i] = sampling_probs[i] * view_selection_priors[i];
	}
	TransformPDFToCDF(sampling_probs, num_images - 1);
	Rng rv(h.seed, (uint32_t)center, rng_site(phase, iter, SUB_VIEW));
	for (int sample = 0; sample < 15; ++sample) {
		const float rand_prob = rv.uniform() - FLT_EPSILON;
		for (int image_id = 0; image_id < num_images - 1; ++image_id) {
			const float prob = sampling_probs[image_id];
			if (prob > rand_prob) { view_weights[image_id] += 1; break; }
		}
	}
	*temp_selected_views = 0;
	*weight_norm = 0;
	for (int i = 0; i < num_images - 1; ++i) {
		if (view_weights[i] > 0) {
			setBit(temp_selected_views, i);
			*weight_norm += view_weights[i];
		}
	}
}

// APD.cu:2010-2141 (use_edge branch), 2462-2567, 2725-2737.  Neighbour planes/costs are read
// from the pre-launch snapshot (h.planes_snap / h.costs_snap).
void CheckerboardPropagationStrong_px(Ctx& h, const int2 p, const int iter) {
	const int width = h.width, height = h.height;
	const float4* plane_hypotheses = h.planes_snap.data();
	const float* costs = h.costs_snap.data();
	const PatchMatchParams& params = h.params;
	const Camera* cameras = h.cameras;
	int num_images = params.num_images;
	const int center = p.y * width + p.x;

	float cost_array[8][32];
	for (int a = 0; a < 8; ++a) for (int b = 0; b < 32; ++b) cost_array[a][b] = 0.0f;
	cost_array[0][0] = 2.0f;   // `= { 2.0f }` initialises one element (APD.cu:2032)
	bool flag[8] = { false };
	int positions[8] = { 0 };

	// the legacy ACMH sampling (use_edge == false, APD.cu:2142-2460) is not restated
	{
		const int dir[EDGE_NEIGH_NUM][2] = { {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, 1}, {-1, 1}, {1, -1} };
		const short2* edge_neigh = &h.edge_neigh[(size_t)center * EDGE_NEIGH_NUM];
		const uint8_t* edge = h.edge.data();
		const float max_edge_dist = ORA_MAX(height, width) / 30.0f;
		const int min_step_len = 2;
		for (int dir_index = 0; dir_index < EDGE_NEIGH_NUM; ++dir_index) {
			const int dx = dir[dir_index][0], dy = dir[dir_index][1];
			const int sx = 5 * dx, sy = 5 * dy;
			short2 edge_pt = edge_neigh[dir_index];
			const double ex = (double)(edge_pt.x - p.x), ey = (double)(edge_pt.y - p.y);
			float dist = (float)std::sqrt(ex * ex + ey * ey);   // std::pow(int,2) -> double (:2054)
			if (dir_index >= 4) dist = (float)((double)dist / std::sqrt(2.0));
			if (edge[center]) {
				dist = 11 * min_step_len;
			} else if (/* !edge_pt.x == -1 is always false (:2059) */ edge_pt.y == -1 || dist >= max_edge_dist) {
				dist = max_edge_dist;
				if (dir_index >= 4) dist = (float)((double)dist / std::sqrt(2.0));
			}
			int step_num = ORA_MIN(ORA_MAX(11, (int)(1.0f * dist / min_step_len)), 22);
			int step_len = ORA_MAX((int)(1.0f * dist / step_num), min_step_len);
			if (dir_index < 4 && step_len % 2 == 1) step_len -= 1;

			int2 min_cost_pt = make_int2(0, 0);
			float min_cost = FLT_MAX;
			for (int step = 0; step < step_num; ++step) {
				int fx = 0, fy = 0;
				if (dir_index > 4) { if (dir_index % 2) fx = dx; else fy = dy; }
				const int2 temp_pt = make_int2(p.x + sx + step * step_len * dx + fx, p.y + sy + step * step_len * dy + fy);
				if (!(temp_pt.x >= 0 && temp_pt.y >= 0 && temp_pt.x < width && temp_pt.y < height)) continue;
				const int pc = temp_pt.x + temp_pt.y * width;
				if (min_cost > costs[pc]) { min_cost_pt = temp_pt; min_cost = costs[pc]; }
			}
			if (min_cost < FLT_MAX) {
				flag[dir_index] = true;
				positions[dir_index] = min_cost_pt.x + min_cost_pt.y * width;
				ComputeMultiViewCostVectorOld(p, plane_hypotheses[positions[dir_index]], cost_array[dir_index], h);
			}
		}
		if (!edge[center]) {
			const float good_threshold = 0.8f * dvp_expf((iter) * (iter) / (-90.0f));
			const float bad_threshold = 1.2f;
			for (int dir_index = 0; dir_index < EDGE_NEIGH_NUM; ++dir_index) {
				const int dx = dir[dir_index][0], dy = dir[dir_index][1];
				const int sx = 5 * dx, sy = 5 * dy;
				bool hasResBefore = flag[dir_index];
				float temp_cost_array[32] = { 2.0f };
				int temp_position;
				int2 min_cost_pt = make_int2(0, 0);
				float min_cost = FLT_MAX;
				for (int step = 0; step < 11; ++step) {
					int fx = 0, fy = 0;
					if (dir_index > 4) { if (dir_index % 2) fx = dx; else fy = dy; }
					const int2 temp_pt = make_int2(p.x + sx + step * min_step_len * dx + fx, p.y + sy + step * min_step_len * dy + fy);
					if (!(temp_pt.x >= 0 && temp_pt.y >= 0 && temp_pt.x < width && temp_pt.y < height)) continue;
					const int pc = temp_pt.x + temp_pt.y * width;
					if (min_cost > costs[pc]) { min_cost_pt = temp_pt; min_cost = costs[pc]; }
				}
				if (min_cost < FLT_MAX) {
					flag[dir_index] = true;
					temp_position = min_cost_pt.x + min_cost_pt.y * width;
					ComputeMultiViewCostVectorOld(p, plane_hypotheses[temp_position], temp_cost_array, h);
					int good_num[2] = { 0, 0 }, bad_num[2] = { 0, 0 };
					for (int i = 0; i < 2; i++)
						for (int j = 0; j < num_images - 1; j++) {
							float val = (i == 0 ? cost_array[dir_index][j] : temp_cost_array[j]);
							if (val < good_threshold) good_num[i]++;
							if (val > bad_threshold) bad_num[i]++;
						}
					if (!hasResBefore || good_num[1] > good_num[0] || (good_num[1] == good_num[0] && bad_num[1] < bad_num[0])) {
						positions[dir_index] = temp_position;
						for (int j = 0; j < num_images - 1; j++) cost_array[dir_index][j] = temp_cost_array[j];
					}
				}
			}
		}
	}

	// Multi-hypothesis Joint View Selection (APD.cu:2462-2530)
	uint8_t* view_weights = &h.view_weight[(size_t)center * MAX_IMAGES];
	for (int i = 0; i < MAX_IMAGES; ++i) view_weights[i] = 0;
	float view_selection_priors[32] = { 0.0f };
	// guards are flag[0],flag[2],flag[4],flag[6] (an ACMH leftover, :2471); center+width on the
	// bottom row reads the zeroed tail of selected_views (the reference reads out of bounds).
	int neighbor_positions[4] = { center - width, center + width, center - 1, center + 1 };
	for (int i = 0; i < 4; ++i) {
		if (flag[2 * i]) {
			for (int j = 0; j < num_images - 1; ++j) {
				if (isSet(h.selected_views[neighbor_positions[i]], j) == 1) view_selection_priors[j] += 0.9f;
				else view_selection_priors[j] += 0.1f;
			}
		}
	}
	uint32_t temp_selected_views = 0;
	float weight_norm = 0;
	JointViewSelection(h, center, iter, PH_STRONG, cost_array, view_selection_priors, view_weights, &temp_selected_views, &weight_norm);

	float final_costs[8] = { 0.0f };
	for (int i = 0; i < 8; ++i) {
		for (int j = 0; j < num_images - 1; ++j)
			if (view_weights[j] > 0) final_costs[i] += view_weights[j] * cost_array[i][j];
		final_costs[i] /= weight_norm;
	}
	const int min_cost_idx = FindMinCostIndex(final_costs, 8);

	float cost_vector_now[32] = { 2.0f };
	ComputeMultiViewCostVectorOld(p, plane_hypotheses[center], cost_vector_now, h);
	float cost_now = 0.0f;
	for (int i = 0; i < num_images - 1; ++i) cost_now += view_weights[i] * cost_vector_now[i];
	cost_now /= weight_norm;
	float costs_center = cost_now;   // costs[center] = cost_now (:2554)
	float depth_now = ComputeDepthfromPlaneHypothesis(cameras[0], plane_hypotheses[center], p);
	float4 plane_hypotheses_now = plane_hypotheses[center];

	if (flag[min_cost_idx]) {
		float depth_before = ComputeDepthfromPlaneHypothesis(cameras[0], plane_hypotheses[positions[min_cost_idx]], p);
		if (depth_before >= params.depth_min && depth_before <= params.depth_max && final_costs[min_cost_idx] < cost_now) {
			depth_now = depth_before;
			plane_hypotheses_now = plane_hypotheses[positions[min_cost_idx]];
			cost_now = final_costs[min_cost_idx];
			h.selected_views[center] = temp_selected_views;
		}
	}

	PlaneHypothesisRefinementStrong(&plane_hypotheses_now, &depth_now, &cost_now, iter, view_weights, weight_norm, p, h);

	if (params.state == REFINE_INIT) {
		if (cost_now < costs_center - 0.1) {   // double compare (:2728)
			costs_center = cost_now;
			h.planes[center] = plane_hypotheses_now;
		}
	} else {
		costs_center = cost_now;
		h.planes[center] = plane_hypotheses_now;
	}
	h.costs[center] = costs_center;
}

// APD.cu:3167-3182
void GetDepthandNormal_px(Ctx& h, const int2 p) {
	const int center = p.y * h.width + p.x;
	h.planes[center].w = ComputeDepthfromPlaneHypothesis(h.cameras[0], h.planes[center], p);
	h.planes[center] = TransformNormal(h.cameras[0], h.planes[center]);
}

// APD.cu:3184-3294
void CheckerboardFilterStrong_px(Ctx& h, const int2 p) {
	int width = h.width, height = h.height;
	float4* pl = h.planes.data();
	const uint8_t* wi = h.weak_info.data();
	const int center = p.y * width + p.x;
	float filter[21];
	int index = 0;
	filter[index++] = pl[center].w;
	const int left = center - 1, leftleft = center - 3;
	const int up = center - width, upup = center - 3 * width;
	const int down = center + width, downdown = center + 3 * width;
	const int right = center + 1, rightright = center + 3;
	if (h.costs[center] < 0.001f) return;
	if (p.y > 0 && wi[up] == STRONG) filter[index++] = pl[up].w;
	if (p.y > 2 && wi[upup] == STRONG) filter[index++] = pl[upup].w;
	if (p.y > 4 && wi[upup - width * 2] == STRONG) filter[index++] = pl[upup - width * 2].w;
	if (p.y < height - 1 && wi[down] == STRONG) filter[index++] = pl[down].w;
	if (p.y < height - 3 && wi[downdown] == STRONG) filter[index++] = pl[downdown].w;
	if (p.y < height - 5 && wi[downdown + width * 2] == STRONG) filter[index++] = pl[downdown + width * 2].w;
	if (p.x > 0 && wi[left] == STRONG) filter[index++] = pl[left].w;
	if (p.x > 2 && wi[leftleft] == STRONG) filter[index++] = pl[leftleft].w;
	if (p.x > 4 && wi[leftleft - 2] == STRONG) filter[index++] = pl[leftleft - 2].w;
	if (p.x < width - 1 && wi[right] == STRONG) filter[index++] = pl[right].w;
	if (p.x < width - 3 && wi[rightright] == STRONG) filter[index++] = pl[rightright].w;
	if (p.x < width - 5 && wi[rightright + 2] == STRONG) filter[index++] = pl[rightright + 2].w;
	if (p.y > 0 && p.x < width - 2 && wi[up + 2] == STRONG) filter[index++] = pl[up + 2].w;
	if (p.y < height - 1 && p.x < width - 2 && wi[down + 2] == STRONG) filter[index++] = pl[down + 2].w;
	if (p.y > 0 && p.x > 1 && wi[up - 2] == STRONG) filter[index++] = pl[up - 2].w;
	if (p.y < height - 1 && p.x > 1 && wi[down - 2] == STRONG) filter[index++] = pl[down - 2].w;
	if (p.x > 0 && p.y > 2 && wi[left - width * 2] == STRONG) filter[index++] = pl[left - width * 2].w;
	if (p.x < width - 1 && p.y > 2 && wi[right - width * 2] == STRONG) filter[index++] = pl[right - width * 2].w;
	if (p.x > 0 && p.y < height - 2 && wi[left + width * 2] == STRONG) filter[index++] = pl[left + width * 2].w;
	if (p.x < width - 1 && p.y < height - 2 && wi[right + width * 2] == STRONG) filter[index++] = pl[right + width * 2].w;
	sort_small(filter, index);
	int median_index = index / 2;
	if (index % 2 == 0) pl[center].w = (filter[median_index - 1] + filter[median_index]) / 2;
	else pl[center].w = filter[median_index];
}

// shared prologue of DepthToWeak / LocalRefine (APD.cu:3928-3960, 4076-4108)
static bool SweepPrologue(Ctx& h, const int2 point, const float4 origin_plane, float origin_depth,
	float* cost_now, float* base_line, float* weight_normal, int* valid_neighbour) {
	const Camera* cameras = h.cameras;
	const int center = point.x + point.y * h.width;
	const uint8_t* view_weight = &h.view_weight[(size_t)MAX_IMAGES * center];
	*cost_now = 0.0f; *base_line = 0; *valid_neighbour = 0; *weight_normal = 0.0f;
	for (int src_index = 1; src_index < h.params.num_images; ++src_index) {
		int view_index = src_index - 1;
		if (isSet(h.selected_views[center], view_index)) {
			float4 temp = origin_plane;
			temp.w = GetDistance2Origin(cameras[0], point, origin_depth, temp);
			float temp_cost = ComputeBilateralNCCOld(point, src_index, temp, h);
			if (h.params.geom_consistency) temp_cost += h.params.geom_factor * ComputeGeomConsistencyCost(point, src_index, temp, h);
			*cost_now += (temp_cost * view_weight[view_index]);
			*weight_normal += view_weight[view_index];
			float c_dist[3];
			c_dist[0] = cameras[0].c[0] - cameras[src_index].c[0];
			c_dist[1] = cameras[0].c[1] - cameras[src_index].c[1];
			c_dist[2] = cameras[0].c[2] - cameras[src_index].c[2];
			double temp_val = c_dist[0] * c_dist[0] + c_dist[1] * c_dist[1] + c_dist[2] * c_dist[2];   // float expr
			*base_line += sqrtf((float)temp_val);
			(*valid_neighbour)++;
		}
	}
	return true;
}

// APD.cu:3892-4051
void DepthToWeak_px(Ctx& h, const int2 point) {
	const int width = h.width, height = h.height;
	const int min_margin = 6;
	const int center = point.x + point.y * width;
	if (h.params.use_radius && h.radius[center] == 0) h.radius[center] = h.params.strong_radius;
	if (point.x < min_margin || point.y < min_margin || point.x >= width - min_margin || point.y >= height - min_margin) {
		h.weak_info[center] = UNKNOWN;
		return;
	}
	const Camera* cameras = h.cameras;
	const int num_images = h.params.num_images;
	const uint8_t* view_weight = &h.view_weight[(size_t)MAX_IMAGES * center];
	float4 origin = TransformNormal2RefCam(cameras[0], h.planes[center]);
	float origin_depth = origin.w;
	if (origin_depth == 0) { h.weak_info[center] = UNKNOWN; return; }
	float cost_now, base_line, weight_normal;
	int valid_neighbour;
	SweepPrologue(h, point, origin, origin_depth, &cost_now, &base_line, &weight_normal, &valid_neighbour);
	if (valid_neighbour == 0) { h.weak_info[center] = UNKNOWN; return; }
	cost_now /= weight_normal;
	base_line /= valid_neighbour;
	float disp = cameras[0].K[0] * base_line / origin_depth;
	const int radius = 30;
	const int p_costs_size = 2 * radius + 1;
	float p_costs[p_costs_size];
	for (int p_disp = -radius; p_disp <= radius; p_disp += 1) {
		float p_depth = cameras[0].K[0] * base_line / (disp + p_disp);
		if (p_depth < h.params.depth_min || p_depth > h.params.depth_max) { p_costs[p_disp + radius] = 2.0f; continue; }
		float4 temp = origin;
		temp.w = GetDistance2Origin(cameras[0], point, p_depth, temp);
		float p_cost = 0.0f;
		for (int src_index = 1; src_index < num_images; ++src_index) {
			int view_index = src_index - 1;
			float temp_cost = 0.0f;
			if (isSet(h.selected_views[center], view_index)) {
				temp_cost += ComputeBilateralNCCOld(point, src_index, temp, h);
				if (h.params.geom_consistency) temp_cost += h.params.geom_factor * ComputeGeomConsistencyCost(point, src_index, temp, h);
				p_cost += (temp_cost * view_weight[view_index]);
			}
		}
		p_cost /= weight_normal;
		p_costs[p_disp + radius] = ORA_MIN(2.0f, p_cost);
	}
	bool is_peak[p_costs_size];
	for (int i = 0; i < p_costs_size; ++i) is_peak[i] = false;
	int peak_count = 0, min_peak = 0;
	float min_cost = 2.0f;
	for (int i = 2; i < p_costs_size - 2; ++i) {
		if (p_costs[i - 1] > p_costs[i] && p_costs[i + 1] > p_costs[i]) {
			is_peak[i] = true;
			peak_count++;
			if (p_costs[i] < min_cost) { min_peak = i; min_cost = p_costs[i]; }
		}
	}
	if (std::abs(min_peak - radius) > h.params.weak_peak_radius || p_costs[min_peak] > 0.5f) { h.weak_info[center] = WEAK; return; }
	if (peak_count == 1) {
		h.weak_info[center] = (p_costs[min_peak] <= 0.15f) ? STRONG : WEAK;
		return;
	}
	float var = 0.0f;
	for (int i = 2; i < p_costs_size - 2; ++i) {
		if (is_peak[i] && i != min_peak) {
			float dist = p_costs[i] - min_cost;
			var += dist * dist;
		}
	}
	var = sqrtf(var);
	var /= (peak_count - 1);
	h.weak_info[center] = (var > 0.2f) ? STRONG : WEAK;
}

// APD.cu:4053-4139
void LocalRefine_px(Ctx& h, const int2 point) {
	const int center = point.x + point.y * h.width;
	const Camera* cameras = h.cameras;
	const int num_images = h.params.num_images;
	const uint8_t* view_weight = &h.view_weight[(size_t)MAX_IMAGES * center];
	float4 origin = TransformNormal2RefCam(cameras[0], h.planes[center]);
	float origin_depth = origin.w;
	if (origin_depth == 0) return;
	float cost_now, base_line, weight_normal;
	int valid_neighbour;
	SweepPrologue(h, point, origin, origin_depth, &cost_now, &base_line, &weight_normal, &valid_neighbour);
	if (weight_normal == 0 || valid_neighbour == 0) return;
	cost_now /= weight_normal;
	base_line /= valid_neighbour;
	float disp = cameras[0].K[0] * base_line / origin_depth;
	const int radius = 5;
	float min_cost = 2.0f;
	float best_depth = origin_depth;
	for (int p_disp = -radius; p_disp <= radius; ++p_disp) {
		float p_depth = cameras[0].K[0] * base_line / (disp + p_disp);
		if (p_depth < h.params.depth_min || p_depth > h.params.depth_max) continue;
		float4 temp = origin;
		temp.w = GetDistance2Origin(cameras[0], point, p_depth, temp);
		float temp_cost = 0.0f;
		for (int src_index = 1; src_index < num_images; ++src_index) {
			int view_index = src_index - 1;
			if (isSet(h.selected_views[center], view_index)) {
				temp_cost += (ComputeBilateralNCCOld(point, src_index, temp, h) * view_weight[view_index]);
				if (h.params.geom_consistency)
					temp_cost += (h.params.geom_factor * ComputeGeomConsistencyCost(point, src_index, temp, h) * view_weight[view_index]);
			}
		}
		temp_cost /= weight_normal;
		if (temp_cost < min_cost) { min_cost = temp_cost; best_depth = p_depth; }
	}
	if (cost_now - min_cost > 0.1) h.planes[center].w = best_depth;   // double compare (:4136)
}

}  // namespace ora
