// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see ora_common.h).
// Weak-pixel path: prior preparation, anchor search, RANSAC fit plane, weak update.
// APD.cu:790-833, 1897-2008, 2739-3089, 3330-3890, 4159-4404.
#include "ora_core.h"
#include "ora_kernels.h"

namespace ora {

static const double kPI = 3.14159265358979323846;   // APD.h:6

// ---- GenEdgeInform (APD.cu:3731-3890) ----------------------------------------------------------
struct SectorPoint { int i, j; double angle; double weight; };   // APD.cu:790-794

static double calculateAngle(int i, int j) {   // APD.cu:797-804
	double angle = std::atan2((double)j, (double)i);
	double deg = angle * (180.0 / kPI);
	if (deg < 0) deg += 360.0;
	return deg;
}
static int getRegion(double angle) {            // APD.cu:807-821
	for (int r = 0; r < 12; ++r)
		if (angle >= 30.0 * r && angle < 30.0 * (r + 1)) return r;
	return -1;
}
static void bubbleSort(SectorPoint* pts, int n) {   // APD.cu:823-833 (descending weight, stable)
	for (int i = 0; i < n - 1; ++i)
		for (int j = 0; j < n - 1 - i; ++j)
			if (pts[j].weight < pts[j + 1].weight) { SectorPoint t = pts[j]; pts[j] = pts[j + 1]; pts[j + 1] = t; }
}

void GenEdgeInform_px(Ctx& h, const int2 point) {
	const int width = h.width, height = h.height;
	const int center = point.x + point.y * width;
	const int S = h.num_images - 1;
	const float* ref_image = h.images[0].data();

	for (int src_idx = 1; src_idx < h.params.num_images; ++src_idx) {
		int radius = h.params.weak_radius;
		SectorPoint regions[12][20];
		int regionCounts[12] = { 0 };
		// regions[r][0] of an empty sector is read uninitialised by the reference (:3783);
		// defined here as the zero point (offset (0,0), weight 0).
		for (auto& r : regions) r[0] = SectorPoint{0, 0, 0.0, 0.0};
		const float ref_center_pix = tex_texel(ref_image, width, height, point.x, point.y);
		for (int i = -radius; i <= radius; i++) {
			for (int j = -radius; j <= radius; j++) {
				if (i == 0 && j == 0) continue;
				double angle = (float)calculateAngle(i, j);   // `float angle = calculateAngle(i, j)` (:3759)
				const int2 ref_pt = make_int2(point.x + i, point.y + j);
				if (ref_pt.x >= 0 && ref_pt.x < width && ref_pt.y >= 0 && ref_pt.y < height) {
					int nei_center = ref_pt.x + ref_pt.y * width;
					if (isSet(h.selected_views[nei_center], src_idx - 1) == 1) {
						const float ref_pix = tex_texel(ref_image, width, height, ref_pt.x, ref_pt.y);
						float weight = ComputeBilateralWeight_YZL((float)i, (float)j, ref_pix, ref_center_pix, h.params.sigma_spatial, h.params.sigma_color);
						SectorPoint sp{i, j, angle, weight};
						int region = getRegion(angle);
						if (region >= 0 && regionCounts[region] < 20) regions[region][regionCounts[region]++] = sp;
					}
				}
			}
		}
		SectorPoint min_regions[12];
		for (int i = 0; i < 12; ++i) bubbleSort(regions[i], regionCounts[i]);
		for (int i = 0; i < 12; ++i) min_regions[i] = regions[i][0];
		bubbleSort(min_regions, 12);
		int ind = src_idx - 1;
		for (int k = 0; k < LAB_BOUNDARY_NUM; k++) {
			short2& c = h.candidate[((size_t)center * S + ind) * LAB_BOUNDARY_NUM + k];
			c.x = (short)min_regions[k].i;
			c.y = (short)min_regions[k].j;
		}
	}

	const uint8_t* edge = h.edge.data();
	const int dir[EDGE_NEIGH_NUM][2] = { {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, 1}, {-1, 1}, {1, -1} };
	if (h.params.use_edge) {
		short2* edge_neigh = &h.edge_neigh[(size_t)center * EDGE_NEIGH_NUM];
		for (int i = 0; i < EDGE_NEIGH_NUM; i++) {
			edge_neigh[i] = make_short2(-1, -1);
			int dx = dir[i][0], dy = dir[i][1];
			int nx = point.x + dx, ny = point.y + dy;
			while (true) {
				if (nx < 0 || nx >= width || ny < 0 || ny >= height) break;
				if (edge[nx + ny * width]) { edge_neigh[i] = make_short2(nx, ny); break; }
				nx += dx; ny += dy;
			}
		}
		if (h.weak_info[center] == WEAK) {
			int radius = h.params.strong_radius;
			int edge_pix = 0, tot_pix = 0;
			for (int i = -radius; i <= radius; i++)
				for (int j = -radius; j <= radius; j++) {
					int nx = point.x + i, ny = point.y + j;
					if (nx < 0 || nx >= width || ny < 0 || ny >= height) continue;
					if (edge[ny * width + nx]) edge_pix++;
					tot_pix++;
				}
			float density = 1.0f * edge_pix / tot_pix;
			// reference: 1.0f / (1.0f + exp(-25.0 * (density - 0.35))) in double (:3844); restated in
			// binary32 with dvp_expf (documented deviation, ~1e-7 relative on a probability).
			if (literal_mode()) h.complex_[h.neighbours_map[center]] = (float)(1.0f / (1.0f + exp(-25.0 * (density - 0.35))));   // APD.cu:3844 in double
			else h.complex_[h.neighbours_map[center]] = 1.0f / (1.0f + dvp_expf(-25.0f * (density - 0.35f)));
		}
		if (h.params.state == REFINE_INIT && h.params.use_detail && edge[center]) {
			if (h.weak_info[center] != STRONG) h.weak_info[center] = UNKNOWN;
		}
	}

	if (h.params.use_label && h.weak_info[center] == WEAK) {
		const unsigned laboff = h.neighbours_map[center] * LAB_BOUNDARY_NUM;
		const int* label_mask = h.label.data();
		short2* lab_bound = &h.label_boundary[laboff];
		int center_label = label_mask[center];
		if (center_label > 0) {
			for (int i = 0; i < LAB_BOUNDARY_NUM; i++) {
				int dx = dir[i][0], dy = dir[i][1];
				int nx = point.x + dx, ny = point.y + dy;
				int last_x = -1, last_y = -1;
				while (true) {
					if (nx < 0 || nx >= width || ny < 0 || ny >= height) break;
					int next_label = label_mask[nx + ny * width];
					if (next_label == center_label) { last_x = nx; last_y = ny; }
					else if (next_label == -1) break;
					nx += dx; ny += dy;
				}
				lab_bound[i] = make_short2(last_x, last_y);
			}
		}
		if (h.params.state == REFINE_INIT && h.params.use_detail && label_mask[center] == 0) {
			if (h.weak_info[center] != STRONG) h.weak_info[center] = UNKNOWN;
		}
	}
}

// ---- FindNearestStrongPoint (APD.cu:4159-4193) -------------------------------------------------
void FindNearestStrongPoint_px(Ctx& h, const int2 point) {
	const int width = h.width, height = h.height;
	const int center = point.x + point.y * width;
	h.weak_nearest_strong[center] = make_short2(-1, -1);
	if (h.weak_info[center] != WEAK) return;
	const int max_radius = 100;
	for (int radius = 0; radius <= max_radius; ++radius)
		for (int x = -radius; x <= radius; ++x)
			for (int y = -radius; y <= radius; ++y) {
				if (std::abs(x) != radius && std::abs(y) != radius) continue;
				const int2 np = make_int2(point.x + x, point.y + y);
				if (np.x < 0 || np.y < 0 || np.x >= width || np.y >= height) continue;
				if (h.weak_info[np.x + np.y * width] == STRONG) {
					h.weak_nearest_strong[center] = make_short2(np.x, np.y);
					return;
				}
			}
}

// ---- GenNeighbours (APD.cu:3330-3711) -----------------------------------------------------------
void GenNeighbours_px(Ctx& h, const int2 point) {
	const int width = h.width, height = h.height;
	const unsigned center = point.x + point.y * width;
	const uint8_t* weak_info = h.weak_info.data();
	if (weak_info[center] != WEAK) return;

	const int max_pt_num = 160;
	const int min_margin = 6;
	const PatchMatchParams& params = h.params;
	const float depth_diff = params.depth_max - params.depth_min;
	const short2* weak_nearest_strong = h.weak_nearest_strong.data();
	const Camera& camera = h.cameras[0];
	const unsigned offset = h.neighbours_map[center] * NEIGHBOUR_NUM;
	const float4* plane_hypotheses = h.planes.data();
	const int* label_mask = h.label.data();
	short2* neighbours = &h.neighbours[offset];
	uint8_t* weak_reliable = &h.weak_reliable[center];
	Rng r_limit(h.seed, center, rng_site(PH_NEIGHBOURS, 0, SUB_LIMIT));
	Rng r_search(h.seed, center, rng_site(PH_NEIGHBOURS, 0, SUB_SEARCH));
	Rng r_ransac(h.seed, center, rng_site(PH_NEIGHBOURS, 0, SUB_RANSAC));

	for (int i = 0; i < NEIGHBOUR_NUM; ++i) neighbours[i] = make_short2(-1, -1);
	neighbours[0] = make_short2(point.x, point.y);
	short2 strong_points[max_pt_num];
	bool dir_valid[max_pt_num];
	for (int i = 0; i < max_pt_num; ++i) { strong_points[i] = make_short2(-1, -1); dir_valid[i] = false; }
	int origin_direction_index = -1;
	int strong_point_size = 0;

	const int rotate_time = params.rotate_time;
	const float angle = 45.0f / rotate_time;
	const float cos_angle_rot = (float)std::cos(angle * kPI / 180.f);
	const float sin_angle_rot = (float)std::sin(angle * kPI / 180.f);
	const float threshhold = (float)std::cos((angle / 2.0f) * kPI / 180.0f);
	const int shift_range = ORA_MAX((int)(std::tan((angle / 2.0f) * kPI / 180.0f) * 20), 1);
	const float ransac_threshold = params.ransac_threshold;

	bool edge_limit = false;
	if (params.use_limit) {
		edge_limit = true;
		if (params.use_edge) {
			float complex_val = h.complex_[h.neighbours_map[center]];
			const float rand_prob = r_limit.uniform() - FLT_EPSILON;
			if (rand_prob < complex_val) edge_limit = false;
		}
	}

	for (int odx = -1; odx <= 1; ++odx) {
		for (int ody = -1; ody <= 1; ++ody) {
			if (odx == 0 && ody == 0) continue;
			float2 origin_direction = make_float2((float)odx, (float)ody);
			NormalizeVec2(&origin_direction);
			origin_direction_index++;
			for (int rotate_iter = 0; rotate_iter < rotate_time; ++rotate_iter) {
				int dir_index = origin_direction_index * 4 + rotate_iter;
				for (int radius = 2; radius <= MAX_SEARCH_RADIUS; radius = ORA_MIN(radius * 2, radius + 25)) {
					float2 test_pt = make_float2(point.x + origin_direction.x * radius, point.y + origin_direction.y * radius);
					if (test_pt.x < 0 || test_pt.y < 0 || test_pt.x >= width || test_pt.y >= height) break;
					for (int radius_iter = 0; radius_iter < 4; ++radius_iter) {
						// (curand()%2==0 ? 1 : -1) * curand() % shift_range — unsigned arithmetic (:3410);
						// first draw = sign, second draw = magnitude.
						uint32_t sgx = (r_search.next() % 2 == 0) ? 1u : 0xFFFFFFFFu;
						int rand_x_shift = (int)((sgx * r_search.next()) % (uint32_t)shift_range);
						uint32_t sgy = (r_search.next() % 2 == 0) ? 1u : 0xFFFFFFFFu;
						int rand_y_shift = (int)((sgy * r_search.next()) % (uint32_t)shift_range);
						float2 direction = make_float2(origin_direction.x * 20 + rand_x_shift, origin_direction.y * 20 + rand_y_shift);
						NormalizeVec2(&direction);
						short2 neighbour_pt = make_short2((int)(point.x + direction.x * radius), (int)(point.y + direction.y * radius));
						if (neighbour_pt.x < min_margin || neighbour_pt.y < min_margin || neighbour_pt.x >= width - min_margin || neighbour_pt.y >= height - min_margin) continue;
						int npc = neighbour_pt.x + neighbour_pt.y * width;
						if (weak_info[npc] != STRONG) {
							neighbour_pt = weak_nearest_strong[npc];
							if (neighbour_pt.x == -1 || neighbour_pt.y == -1) continue;
							npc = neighbour_pt.x + neighbour_pt.y * width;
						}
						bool has_same_pt = false;
						for (int k = 0; k < dir_index; k++)
							if (strong_points[k].x == neighbour_pt.x && strong_points[k].y == neighbour_pt.y) { has_same_pt = true; break; }
						if (has_same_pt) continue;
						float2 test_direction = make_float2((float)(neighbour_pt.x - point.x), (float)(neighbour_pt.y - point.y));
						NormalizeVec2(&test_direction);
						float cos_a = Vec2DotVec2(test_direction, origin_direction);
						if (cos_a > threshhold && (!edge_limit || !BresenhamLine(point, make_int2(neighbour_pt.x, neighbour_pt.y), h))) {
							strong_points[dir_index] = neighbour_pt;
							dir_valid[dir_index] = true;
							strong_point_size++;
							break;
						}
					}
					if (dir_valid[dir_index]) break;
				}
				{   // rotate
					float2 rd;
					rd.x = origin_direction.x * cos_angle_rot - origin_direction.y * sin_angle_rot;
					rd.y = origin_direction.x * sin_angle_rot + origin_direction.y * cos_angle_rot;
					NormalizeVec2(&rd);
					origin_direction = rd;
				}
			}
		}
	}

	int extend_index = 31;
	if (params.use_label && label_mask[center] > 0) {
		// int array initialised with 0.5 entries that truncate to 0 (APD.cu:3462)
		const int dir[LAB_BOUNDARY_NUM * 2][2] = { {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, 1}, {-1, 1}, {1, -1}, {1, 0}, {0, 1}, {0, 1}, {-1, 0}, {-1, 0}, {0, -1}, {0, -1}, {1, 0} };
		const int laboff = h.neighbours_map[center] * LAB_BOUNDARY_NUM;
		const short2* lab_bound = &h.label_boundary[laboff];
		float bound_dist[LAB_BOUNDARY_NUM * 2] = { 0 };
		int dir_step[LAB_BOUNDARY_NUM * 2] = { 0 };
		for (int i = 0; i < LAB_BOUNDARY_NUM; ++i) {
			short2 bp = lab_bound[i];
			float dist = 0.0f;
			if (bp.x != -1 && bp.y != -1) {
				const double ex = (double)(point.x - bp.x), ey = (double)(point.y - bp.y);
				dist = (float)std::sqrt(ex * ex + ey * ey);
				if (i >= 4) dist = (float)((double)dist / std::sqrt(2.0));
			}
			bound_dist[i] = dist;
			if (i % 2 == 1) {
				// MIN(1, MAX(4*rt-1, (int)(...))) is always 1 (APD.cu:3477)
				int step = 1;
				int opposite_step = 4 * rotate_time - step;
				dir_step[i - 1] = opposite_step;
				dir_step[i] = step;
			}
		}
		const int comb[8][2] = { {3, 5}, {1, 5}, {1, 6}, {2, 6}, {2, 4}, {4, 0}, {7, 0}, {7, 3} };   // APD.cu:3484-3506
		for (int q = 0; q < 8; ++q) {
			dir_step[8 + q] = (dir_step[comb[q][0]] + dir_step[comb[q][1]]) / 2;
			bound_dist[8 + q] = (bound_dist[comb[q][0]] + bound_dist[comb[q][1]]) / 2;
		}
		for (int i = 0; i < LAB_BOUNDARY_NUM * 2; ++i) {
			float dist = bound_dist[i];
			int gap_num = dir_step[i] + 1;
			int step_len = ORA_MAX(1, (int)std::floor(1.0 * dist / gap_num));
			for (int step = 1; step <= dir_step[i]; ++step) {
				short2 neighbour_pt = make_short2(point.x + step * step_len * dir[i][0], point.y + step * step_len * dir[i][1]);
				if (neighbour_pt.x < min_margin || neighbour_pt.y < min_margin || neighbour_pt.x >= width - min_margin || neighbour_pt.y >= height - min_margin) continue;
				int npc = neighbour_pt.x + neighbour_pt.y * width;
				if (weak_info[npc] != STRONG) {
					neighbour_pt = weak_nearest_strong[npc];
					if (neighbour_pt.x == -1 || neighbour_pt.y == -1) continue;
					npc = neighbour_pt.x + neighbour_pt.y * width;
				}
				bool has_same_pt = false;
				for (int k = 0; k <= extend_index; k++)
					if (strong_points[k].x == neighbour_pt.x && strong_points[k].y == neighbour_pt.y) { has_same_pt = true; break; }
				if (has_same_pt) continue;
				if (extend_index + 1 >= max_pt_num) continue;   // array bound (cannot trigger for rotate_time <= 4)
				extend_index++;
				strong_points[extend_index] = neighbour_pt;
				dir_valid[extend_index] = true;
				strong_point_size++;
			}
		}
	}

	if (strong_point_size <= 3) { *weak_reliable = 0; return; }

	float4 best_plane = make_float4(0, 0, 0, 0);
	bool has_valid_plane = false;
	short2 strong_points_valid[max_pt_num];
	float3 strong_points_valid_3d[max_pt_num];
	float3 strong_points_valid_normals[max_pt_num];
	int valid_count = 0;
	float X[3];
	Get3DPoint(camera, point, plane_hypotheses[center].w, X);
	float3 center_point_world = make_float3(X[0], X[1], X[2]);
	for (int i = 0; i < max_pt_num; ++i) {
		strong_points_valid[i] = make_short2(-1, -1);
		if (dir_valid[i]) {
			const short2 sp = strong_points[i];
			int spc = sp.x + sp.y * width;
			strong_points_valid[valid_count] = sp;
			Get3DPoint(camera, sp, plane_hypotheses[spc].w, X);
			strong_points_valid_3d[valid_count] = make_float3(X[0], X[1], X[2]);
			float4 n4 = TransformNormal2RefCam(camera, plane_hypotheses[spc]);
			strong_points_valid_normals[valid_count] = make_float3(n4.x, n4.y, n4.z);
			valid_count++;
		}
	}
	{
		int iteration = 300, max_iter = 200;
		float min_cost = FLT_MAX;
		int max_count = 3;
		bool has_strong_plane = false;
		// symmetric first-evaluation-wins cache of the line tests (APD.cu:3574, 3588-3604): BresenhamLine
		// walks from its second argument with a step limit, so the answer for an unordered pair is the
		// one of the orientation in which the pair was first tested
		static thread_local std::vector<unsigned char> edge_test((size_t)max_pt_num * max_pt_num, 0);
		for (int a = 0; a < valid_count; ++a) std::memset(&edge_test[(size_t)a * max_pt_num], 0, (size_t)valid_count);   // only [0, valid_count)^2 is used
		auto tested = [&](int a, int b) -> unsigned char& { return edge_test[(size_t)a * max_pt_num + b]; };
		while (iteration > 0 && max_iter > 0) {
			max_iter--;
			int a_index = (int)(r_ransac.next() % (uint32_t)valid_count);
			int b_index = (int)(r_ransac.next() % (uint32_t)valid_count);
			int c_index = (int)(r_ransac.next() % (uint32_t)valid_count);
			if (a_index == b_index || b_index == c_index || a_index == c_index) continue;
			if (!PointinTriangle(strong_points_valid[a_index], strong_points_valid[b_index], strong_points_valid[c_index], point)) continue;
			if (edge_limit) {
				if (tested(a_index, b_index) == 0)
					tested(a_index, b_index) = tested(b_index, a_index) = BresenhamLine(strong_points_valid[a_index], strong_points_valid[b_index], h) ? 1 : 2;
				if (tested(b_index, c_index) == 0)
					tested(b_index, c_index) = tested(c_index, b_index) = BresenhamLine(strong_points_valid[b_index], strong_points_valid[c_index], h) ? 1 : 2;
				if (tested(c_index, a_index) == 0)
					tested(c_index, a_index) = tested(a_index, c_index) = BresenhamLine(strong_points_valid[c_index], strong_points_valid[a_index], h) ? 1 : 2;
				if (tested(a_index, b_index) == 1 || tested(b_index, c_index) == 1 || tested(c_index, a_index) == 1) continue;
			}
			const float3& AN = strong_points_valid_normals[a_index];
			const float3& BN = strong_points_valid_normals[a_index];   // a_index three times (:3605-3607): kept
			const float3& CN = strong_points_valid_normals[a_index];
			if (Vec3DotVec3(AN, BN) < 0.9f || Vec3DotVec3(AN, CN) < 0.9f || Vec3DotVec3(BN, CN) < 0.9f) continue;
			const float3& A = strong_points_valid_3d[a_index];
			const float3& B = strong_points_valid_3d[b_index];
			const float3& C = strong_points_valid_3d[c_index];
			float3 A_C = make_float3(A.x - C.x, A.y - C.y, A.z - C.z);
			float3 B_C = make_float3(B.x - C.x, B.y - C.y, B.z - C.z);
			float4 cross_vec;
			cross_vec.x = A_C.y * B_C.z - B_C.y * A_C.z;
			cross_vec.y = -(A_C.x * B_C.z - B_C.x * A_C.z);
			cross_vec.z = A_C.x * B_C.y - B_C.x * A_C.y;
			if ((cross_vec.x == 0 && cross_vec.y == 0 && cross_vec.z == 0) || std::isnan(cross_vec.x) || std::isnan(cross_vec.y) || std::isnan(cross_vec.z)) continue;
			iteration--;
			NormalizeVec3(&cross_vec);
			cross_vec.w = -(cross_vec.x * A.x + cross_vec.y * A.y + cross_vec.z * A.z);
			bool is_strong_plane = true;
			if (params.use_label && label_mask[center] > 0 && fabsf(Vec3DotVec3(AN, cross_vec)) < 0.9f && fabsf(Vec3DotVec3(BN, cross_vec)) < 0.9f && fabsf(Vec3DotVec3(CN, cross_vec)) < 0.9f)
				is_strong_plane = false;
			if (has_strong_plane && !is_strong_plane) continue;
			int temp_count = 0;
			float strong_dist = 0.0f;
			for (int si = 0; si < valid_count; ++si) {
				const float3& tp = strong_points_valid_3d[si];
				const short2& tpos = strong_points_valid[si];
				float factor_x = (tpos.x - camera.K[2]) / camera.K[0];
				float factor_y = (tpos.y - camera.K[5]) / camera.K[4];
				float fit_depth = -cross_vec.w / (cross_vec.x * factor_x + cross_vec.y * factor_y + cross_vec.z);
				float distance = fabsf(fit_depth - tp.z);
				if (distance / depth_diff < ransac_threshold) { temp_count++; strong_dist += distance; }
			}
			if (temp_count < 6) continue;
			float factor_x = (point.x - camera.K[2]) / camera.K[0];
			float factor_y = (point.y - camera.K[5]) / camera.K[4];
			float fit_depth = -cross_vec.w / (cross_vec.x * factor_x + cross_vec.y * factor_y + cross_vec.z);
			const float center_distance = fabsf(fit_depth - center_point_world.z);
			if (temp_count > max_count || (!has_strong_plane && is_strong_plane)) {
				if (!has_strong_plane && is_strong_plane) has_strong_plane = true;
				best_plane = cross_vec;
				max_count = temp_count;
				min_cost = center_distance;
				has_valid_plane = true;
			} else if (temp_count == max_count) {
				if (center_distance < min_cost) {
					best_plane = cross_vec;
					max_count = temp_count;
					min_cost = center_distance;
				}
			}
		}
	}

	float weight[max_pt_num];
	if (!has_valid_plane) { *weak_reliable = 0; return; }
	for (int i = 0; i < valid_count; ++i) {
		const float3& tp = strong_points_valid_3d[i];
		const short2& tpos = strong_points_valid[i];
		float factor_x = (tpos.x - camera.K[2]) / camera.K[0];
		float factor_y = (tpos.y - camera.K[5]) / camera.K[4];
		float fit_depth = -best_plane.w / (best_plane.x * factor_x + best_plane.y * factor_y + best_plane.z);
		float distance = fabsf(fit_depth - tp.z);
		if (distance / depth_diff >= ransac_threshold) {
			strong_points_valid[i] = make_short2(-1, -1);
			weight[i] = FLT_MAX;
			continue;
		}
		weight[i] = distance;
	}
	sort_small_weighted(strong_points_valid, weight, valid_count);
	for (int i = 1; i < NEIGHBOUR_NUM; ++i) neighbours[i] = strong_points_valid[i - 1];
	*weak_reliable = 1;
}

// APD.cu:3713-3729
void NeigbourUpdate_px(Ctx& h, const int2 point) {
	const int center = point.x + point.y * h.width;
	if (h.weak_info[center] != WEAK) return;
	if (h.weak_reliable[center] != 1) h.weak_info[center] = UNKNOWN;
}

// ---- RANSACToGetFitPlane (APD.cu:4195-4404) -----------------------------------------------------
void RANSACToGetFitPlane_px(Ctx& h, const int2 point, int iter) {
	const int width = h.width;
	const int center = point.x + point.y * width;
	const PatchMatchParams& params = h.params;
	float4* plane_hypotheses = h.planes.data();
	float4* fit = h.fit_planes.data();
	if (h.weak_info[center] != WEAK) { fit[center] = plane_hypotheses[center]; return; }
	const Camera& camera = h.cameras[0];
	Rng r_limit(h.seed, (uint32_t)center, rng_site(PH_RANSAC, iter, SUB_LIMIT));
	Rng r_ransac(h.seed, (uint32_t)center, rng_site(PH_RANSAC, iter, SUB_RANSAC));

	bool edge_limit = false;
	if (params.use_limit) {
		edge_limit = true;
		if (params.use_edge) {
			float complex_val = h.complex_[h.neighbours_map[center]];
			const float rand_prob = r_limit.uniform() - FLT_EPSILON;
			if (rand_prob < complex_val) edge_limit = false;
		}
	}
	short2 strong_points[NEIGHBOUR_NUM - 1];
	float3 strong_points_3d[NEIGHBOUR_NUM - 1];
	float3 strong_points_normals[NEIGHBOUR_NUM - 1];
	int strong_count = 0;
	float X[3];
	for (int i = 1; i < NEIGHBOUR_NUM; ++i) {
		short2 tp = GetNeighbourPoint(point, i, h);
		if (tp.x == -1 || tp.y == -1) continue;
		strong_points[strong_count] = tp;
		const int tc = tp.x + tp.y * width;
		float depth = ComputeDepthfromPlaneHypothesis(camera, plane_hypotheses[tc], make_int2(tp.x, tp.y));
		Get3DPoint(camera, strong_points[strong_count], depth, X);
		strong_points_3d[strong_count] = make_float3(X[0], X[1], X[2]);
		float4 n4 = plane_hypotheses[tc];
		strong_points_normals[strong_count] = make_float3(n4.x, n4.y, n4.z);
		strong_count++;
	}
	if (strong_count < 3) { fit[center] = plane_hypotheses[center]; return; }

	int iteration = 50;
	int use_a = -1, use_b = -1, use_c = -1;
	float min_cost = FLT_MAX;
	float4 best_plane = make_float4(0, 0, 0, 0);
	bool has_best_plane = false;
	unsigned char edge_test[NEIGHBOUR_NUM - 1][NEIGHBOUR_NUM - 1] = {};   // APD.cu:4260: symmetric, first evaluation wins
	while (iteration--) {
		int a_index = (int)(r_ransac.next() % (uint32_t)strong_count);
		int b_index = (int)(r_ransac.next() % (uint32_t)strong_count);
		int c_index = (int)(r_ransac.next() % (uint32_t)strong_count);
		if (a_index == b_index || b_index == c_index || a_index == c_index) continue;
		const float3& AN = strong_points_normals[a_index];
		const float3& BN = strong_points_normals[b_index];
		const float3& CN = strong_points_normals[c_index];
		if (Vec3DotVec3(AN, BN) < 0.9f || Vec3DotVec3(AN, CN) < 0.9f || Vec3DotVec3(BN, CN) < 0.9f) continue;
		if (!PointinTriangle(strong_points[a_index], strong_points[b_index], strong_points[c_index], point)) continue;
		if (edge_limit) {
			if (edge_test[a_index][b_index] == 0)
				edge_test[a_index][b_index] = edge_test[b_index][a_index] = BresenhamLine(strong_points[a_index], strong_points[b_index], h) ? 1 : 2;
			if (edge_test[b_index][c_index] == 0)
				edge_test[b_index][c_index] = edge_test[c_index][b_index] = BresenhamLine(strong_points[b_index], strong_points[c_index], h) ? 1 : 2;
			if (edge_test[c_index][a_index] == 0)
				edge_test[c_index][a_index] = edge_test[a_index][c_index] = BresenhamLine(strong_points[c_index], strong_points[a_index], h) ? 1 : 2;
			if (edge_test[a_index][b_index] == 1 || edge_test[b_index][c_index] == 1 || edge_test[c_index][a_index] == 1) continue;
		}
		const float3& A = strong_points_3d[a_index];
		const float3& B = strong_points_3d[b_index];
		const float3& C = strong_points_3d[c_index];
		float3 A_C = make_float3(A.x - C.x, A.y - C.y, A.z - C.z);
		float3 B_C = make_float3(B.x - C.x, B.y - C.y, B.z - C.z);
		float4 cross_vec;
		cross_vec.x = A_C.y * B_C.z - B_C.y * A_C.z;
		cross_vec.y = -(A_C.x * B_C.z - B_C.x * A_C.z);
		cross_vec.z = A_C.x * B_C.y - B_C.x * A_C.y;
		if ((cross_vec.x == 0 && cross_vec.y == 0 && cross_vec.z == 0) || std::isnan(cross_vec.x) || std::isnan(cross_vec.y) || std::isnan(cross_vec.z)) continue;
		NormalizeVec3(&cross_vec);
		cross_vec.w = -(cross_vec.x * A.x + cross_vec.y * A.y + cross_vec.z * A.z);
		float temp_cost = 0.0f;
		for (int si = 0; si < strong_count; ++si) {
			if (si == a_index || si == b_index || si == c_index) continue;
			const float3& tp = strong_points_3d[si];
			const short2& tpix = strong_points[si];
			float factor_x = (tpix.x - camera.K[2]) / camera.K[0];
			float factor_y = (tpix.y - camera.K[5]) / camera.K[4];
			float fit_depth = -cross_vec.w / (cross_vec.x * factor_x + cross_vec.y * factor_y + cross_vec.z);
			temp_cost += fabsf(fit_depth - tp.z);
		}
		if (temp_cost < min_cost) {
			min_cost = temp_cost;
			best_plane = cross_vec;
			has_best_plane = true;
			// the reference never assigns use_a/b/c_index and then reads strong_points[-1]
			// (APD.cu:4257,4349-4351); defined here as the triangle that produced best_plane.
			use_a = a_index; use_b = b_index; use_c = c_index;
		}
	}

	if (has_best_plane) {
		float depth = ComputeDepthfromPlaneHypothesis(camera, plane_hypotheses[center], point);
		float4 view_direction = GetViewDirection(camera, point, depth);
		float dot_product = best_plane.x * view_direction.x + best_plane.y * view_direction.y + best_plane.z * view_direction.z;
		if (dot_product > 0) {
			best_plane.x = -best_plane.x; best_plane.y = -best_plane.y;
			best_plane.z = -best_plane.z; best_plane.w = -best_plane.w;
		}
		fit[center] = best_plane;
		if (params.use_radius) {
			const short2& A = strong_points[use_a];
			const short2& B = strong_points[use_b];
			const short2& C = strong_points[use_c];
			float a = sqrtf((float)((A.x - B.x) * (A.x - B.x) + (A.y - B.y) * (A.y - B.y)));
			float b = sqrtf((float)((B.x - C.x) * (B.x - C.x) + (B.y - C.y) * (B.y - C.y)));
			float c = sqrtf((float)((C.x - A.x) * (C.x - A.x) + (C.y - A.y) * (C.y - A.y)));
			float pp = (float)((a + b + c) / 2.0);
			float Sa = sqrtf(pp * (pp - a) * (pp - b) * (pp - c));
			// a degenerate triangle gives a NaN area; (int)floor(NaN) is undefined -> defined as 0
			const double rr = std::floor(sqrtf(Sa) / 2.0);
			int radius = (rr == rr) ? (int)rr : 0;
			float A_dis = sqrtf((float)((A.x - point.x) * (A.x - point.x) + (A.y - point.y) * (A.y - point.y)));
			float B_dis = sqrtf((float)((B.x - point.x) * (B.x - point.x) + (B.y - point.y) * (B.y - point.y)));
			float C_dis = sqrtf((float)((C.x - point.x) * (C.x - point.x) + (C.y - point.y) * (C.y - point.y)));
			float min_dis = ORA_MIN(ORA_MIN(A_dis, B_dis), C_dis);
			if (2.5 * min_dis < radius) radius = (int)min_dis;
			if (edge_limit) {
				if (params.use_edge) {
					float min_edge_dist = FLT_MAX;
					const short2* edge_neigh = &h.edge_neigh[(size_t)center * EDGE_NEIGH_NUM];
					for (int d = 0; d < EDGE_NEIGH_NUM; ++d) {
						short2 ep = edge_neigh[d];
						if (ep.x == -1 || ep.y == -1) continue;
						float dist = sqrtf((float)((ep.x - point.x) * (ep.x - point.x) + (ep.y - point.y) * (ep.y - point.y)));
						min_edge_dist = ORA_MIN(min_edge_dist, dist);
					}
					if (min_edge_dist < radius) radius = (int)min_edge_dist;
				}
				if (params.use_label) {
					float min_boundary_dist = FLT_MAX;
					const unsigned laboff = h.neighbours_map[center] * LAB_BOUNDARY_NUM;
					const short2* lab_bound = &h.label_boundary[laboff];
					for (int d = 0; d < LAB_BOUNDARY_NUM; ++d) {
						short2 bp = lab_bound[d];
						if (bp.x == -1 || bp.y == -1) continue;
						const double ex = (double)(point.x - bp.x), ey = (double)(point.y - bp.y);
						float dist = (float)std::sqrt(ex * ex + ey * ey);
						min_boundary_dist = ORA_MIN(min_boundary_dist, dist);
					}
					if (min_boundary_dist < radius) radius = (int)min_boundary_dist;
				}
			}
			if (radius < 0) radius = 0;
			while ((radius << 1) % 5 != 0) radius--;
			h.radius[center] = radius < params.strong_radius ? 0 : radius;
		}
	} else {
		fit[center] = make_float4(0, 0, 0, 0);
		if (params.use_radius) h.radius[center] = params.strong_radius;
	}
}

// ---- weak update (APD.cu:1897-2008, 2739-3089) ---------------------------------------------------
static void PlaneHypothesisRefinementWeak(float4* plane_hypothesis, float* depth, float* cost, int iter,
	const uint8_t* view_weights, const float weight_norm, const int2 p, Ctx& h) {
	float depth_perturbation = 0.02f;
	const Camera* cameras = h.cameras;
	const PatchMatchParams& params = h.params;
	float depth_min = params.depth_min, depth_max = params.depth_max;
	const int center = p.x + p.y * h.width;
	auto weighted = [&](const float4 pl, const float* cost_vector) {
		float temp_cost = 0.0f;
		for (int j = 0; j < params.num_images - 1; ++j) {
			if (view_weights[j] > 0) {
				if (params.geom_consistency) temp_cost += view_weights[j] * (cost_vector[j] + params.geom_factor * ComputeGeomConsistencyCost(p, j + 1, pl, h));
				else temp_cost += view_weights[j] * cost_vector[j];
			}
		}
		return temp_cost / weight_norm;
	};
	if (h.weak_info[center] == WEAK) {
		float4 fitp = h.fit_planes[center];
		if (fitp.x == 0 && fitp.y == 0 && fitp.z == 0) return;
		float cost_vector[32] = { 2.0f };
		ComputeMultiViewCostVectorNew(p, fitp, cost_vector, h);
		float temp_cost = weighted(fitp, cost_vector);
		float depth_before = ComputeDepthfromPlaneHypothesis(cameras[0], fitp, p);
		if (depth_before >= depth_min && depth_before <= depth_max && temp_cost < *cost) {
			*depth = depth_before; *plane_hypothesis = fitp; *cost = temp_cost;
		}
	}
	{
		const uint32_t pix = (uint32_t)center;
		Rng rd(h.seed, pix, rng_site(PH_WEAK, iter, SUB_DEPTH_RAND));
		Rng rn(h.seed, pix, rng_site(PH_WEAK, iter, SUB_NORMAL));
		Rng rp(h.seed, pix, rng_site(PH_WEAK, iter, SUB_DEPTH_PERT));
		float depth_rand = rd.uniform() * (depth_max - depth_min) + depth_min;
		float4 plane_hypothesis_rand = GenerateRandomNormal_YZL(h, cameras[0], p, rn, *depth);
		float depth_perturbed = *depth;
		const float depth_min_perturbed = (1 - depth_perturbation) * depth_perturbed;
		const float depth_max_perturbed = (1 + depth_perturbation) * depth_perturbed;
		depth_perturbed = rp.uniform() * (depth_max_perturbed - depth_min_perturbed) + depth_min_perturbed;
		float4 pert = *plane_hypothesis;   // GeneratePerturbedNormal returns the normalised input (:617-661)
		NormalizeVec3(&pert);
		const int num_planes = 6;
		float depths[num_planes] = { depth_rand, *depth, depth_rand, *depth, *depth, depth_perturbed };
		float4 normals[num_planes] = { *plane_hypothesis, plane_hypothesis_rand, plane_hypothesis_rand, pert, pert, *plane_hypothesis };
		for (int i = 0; i < num_planes; ++i) {
			float cost_vector[32] = { 2.0f };
			float4 temp = normals[i];
			temp.w = GetDistance2Origin(cameras[0], p, depths[i], temp);
			ComputeMultiViewCostVectorNew(p, temp, cost_vector, h);
			float temp_cost = weighted(temp, cost_vector);
			float depth_before = ComputeDepthfromPlaneHypothesis(cameras[0], temp, p);
			if (depth_before >= depth_min && depth_before <= depth_max && temp_cost < *cost) {
				*depth = depth_before; *plane_hypothesis = temp; *cost = temp_cost;
			}
		}
	}
}

void CheckerboardPropagationWeak_px(Ctx& h, const int2 p, const int iter) {
	const int width = h.width;
	float4* plane_hypotheses = h.planes.data();
	const PatchMatchParams& params = h.params;
	const Camera* cameras = h.cameras;
	int num_images = params.num_images;
	const int center = p.y * width + p.x;

	float cost_array[8][32];
	for (int a = 0; a < 8; ++a) for (int b = 0; b < 32; ++b) cost_array[a][b] = 0.0f;
	cost_array[0][0] = 2.0f;
	bool flag[8] = { false };
	int positions[8] = { 0 };
	float4 new_plane_hypothesis[8];
	for (int i = 0; i < 8; ++i) {
		const short2 np = GetNeighbourPoint(p, i + 1, h);
		if (np.x == -1 || np.y == -1 || h.weak_info[np.x + np.y * width] != STRONG) { flag[i] = false; continue; }
		positions[i] = np.x + np.y * width;
		flag[i] = true;
		ComputeMultiViewCostVectorNew(p, plane_hypotheses[positions[i]], cost_array[i], h);
		new_plane_hypothesis[i] = plane_hypotheses[positions[i]];
	}
	uint8_t* view_weights = &h.view_weight[(size_t)center * MAX_IMAGES];
	for (int i = 0; i < MAX_IMAGES; ++i) view_weights[i] = 0;
	float view_selection_priors[32] = { 0.0f };
	for (int i = 0; i < 8; ++i) {
		const short2 np = GetNeighbourPoint(p, i + 1, h);
		if (np.x == -1 || np.y == -1) continue;
		for (int j = 0; j < num_images - 1; ++j) {
			if (isSet(h.selected_views[np.x + np.y * width], j) == 1) view_selection_priors[j] += 0.9f;
			else view_selection_priors[j] += 0.1f;
		}
	}
	uint32_t temp_selected_views = 0;
	float weight_norm = 0;
	JointViewSelection(h, center, iter, PH_WEAK, cost_array, view_selection_priors, view_weights, &temp_selected_views, &weight_norm);

	float final_costs[8] = { 0.0f };
	for (int i = 0; i < 8; ++i) {
		for (int j = 0; j < num_images - 1; ++j) {
			if (view_weights[j] > 0) {
				if (params.geom_consistency) {
					if (flag[i]) final_costs[i] += view_weights[j] * (cost_array[i][j] + params.geom_factor * ComputeGeomConsistencyCost(p, j + 1, plane_hypotheses[positions[i]], h));
					else final_costs[i] += view_weights[j] * (cost_array[i][j] + params.geom_factor * 3.0f);
				} else {
					final_costs[i] += view_weights[j] * cost_array[i][j];
				}
			}
		}
		final_costs[i] /= weight_norm;
	}
	const int min_cost_idx = FindMinCostIndex(final_costs, 8);

	float cost_vector_now[32] = { 2.0f };
	ComputeMultiViewCostVectorNew(p, plane_hypotheses[center], cost_vector_now, h);
	float cost_now = 0.0f;
	for (int i = 0; i < num_images - 1; ++i) {
		if (params.geom_consistency) cost_now += view_weights[i] * (cost_vector_now[i] + params.geom_factor * ComputeGeomConsistencyCost(p, i + 1, plane_hypotheses[center], h));
		else cost_now += view_weights[i] * cost_vector_now[i];
	}
	cost_now /= weight_norm;
	float costs_center = cost_now;
	float depth_now = ComputeDepthfromPlaneHypothesis(cameras[0], plane_hypotheses[center], p);
	float4 plane_hypotheses_now = plane_hypotheses[center];
	if (flag[min_cost_idx]) {
		float depth_before = ComputeDepthfromPlaneHypothesis(cameras[0], new_plane_hypothesis[min_cost_idx], p);
		if (depth_before >= params.depth_min && depth_before <= params.depth_max && final_costs[min_cost_idx] < cost_now) {
			depth_now = depth_before;
			plane_hypotheses_now = new_plane_hypothesis[min_cost_idx];
			cost_now = final_costs[min_cost_idx];
			h.selected_views[center] = temp_selected_views;
		}
	}
	PlaneHypothesisRefinementWeak(&plane_hypotheses_now, &depth_now, &cost_now, iter, view_weights, weight_norm, p, h);
	if (params.state == REFINE_INIT) {
		if (cost_now < costs_center - 0.1) { costs_center = cost_now; plane_hypotheses[center] = plane_hypotheses_now; }
	} else {
		costs_center = cost_now;
		plane_hypotheses[center] = plane_hypotheses_now;
	}
	{   // update cost with old method at the default radius (APD.cu:3072-3088)
		int temp_radius = params.strong_radius;
		if (params.use_radius) { temp_radius = h.radius[center]; h.radius[center] = params.strong_radius; }
		cost_now = 0.0f;
		ComputeMultiViewCostVectorOld(p, plane_hypotheses[center], cost_vector_now, h);
		for (int i = 0; i < num_images - 1; ++i) cost_now += view_weights[i] * cost_vector_now[i];
		cost_now /= weight_norm;
		h.costs[center] = cost_now;
		if (params.use_radius) h.radius[center] = temp_radius;
	}
}

}  // namespace ora
