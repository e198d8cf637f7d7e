// fusion.cpp — RunFusion, ETH variant (/root/reference/APD.cpp:1809-1960): cross-view consistency
// voting on the CPU and binary PLY export.  Sequential and order dependent (the `masks` side
// effects), exactly like the reference.
#include "APD.h"
#include <unordered_map>

static float3 Get3DPointonWorld(const int x, const int y, const float depth, const Camera& camera) {   // APD.cpp:502-523
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
// ProjectCamera (APD.cpp:536-546) is defined in prior.cpp
static float GetAngle(const Vec3f& v1, const Vec3f& v2) {   // APD.cpp:1797-1806
	float dot_product = v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2];
	float angle = acosf(dot_product);
	if (angle != angle) return 0.0f;
	return angle;
}
// RescaleImageAndCamera (APD.cpp:1750-1771) for an 8UC3 image: cv::resize(INTER_LINEAR) per channel
static Mat RescaleColor(const Mat& src, int cols, int rows, Camera& camera) {
	if (cols == src.cols && rows == src.rows) return src.clone();
	const float scale_x = cols / static_cast<float>(src.cols);
	const float scale_y = rows / static_cast<float>(src.rows);
	Mat dst(rows, cols, CV_8UC3);
	for (int ch = 0; ch < 3; ++ch) {
		Mat f(src.rows, src.cols, CV_32FC1);
		for (int r = 0; r < src.rows; ++r)
			for (int c = 0; c < src.cols; ++c) f.at<float>(r, c) = src.data[(size_t)r * src.step + 3 * c + ch];
		Mat g = ResizeLinear(f, cols, rows);
		for (int r = 0; r < rows; ++r)
			for (int c = 0; c < cols; ++c) dst.data[(size_t)r * dst.step + 3 * c + ch] = (uint8_t)std::lrintf(std::min(255.f, std::max(0.f, g.at<float>(r, c))));
	}
	camera.K[0] *= scale_x; camera.K[2] *= scale_x;
	camera.K[4] *= scale_y; camera.K[5] *= scale_y;
	camera.width = cols; camera.height = rows;
	return dst;
}

void RunFusion(const path& dense_folder, const std::vector<Problem>& problems) {
	int num_images = (int)problems.size();
	path image_folder = dense_folder / path("images");
	path cam_folder = dense_folder / path("cams");
	std::vector<Mat> images, depths, normals, masks, weaks;
	std::vector<Camera> cameras;
	std::unordered_map<int, int> imageIdToindexMap;
	for (int i = 0; i < num_images; ++i) {
		const auto& problem = problems[i];
		std::cout << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		imageIdToindexMap.emplace(problem.ref_image_id, i);
		Mat image = ReadImageColor(image_folder / path(ToFormatIndex(problem.ref_image_id) + ".jpg"));
		Camera camera;
		ReadCamera(cam_folder / path(ToFormatIndex(problem.ref_image_id) + "_cam.txt"), camera);
		Mat depth, normal, weak;
		ReadBinMat(problem.result_folder / path("depths.dmb"), depth);
		ReadBinMat(problem.result_folder / path("APD_normals.dmb"), normal);
		ReadBinMat(problem.result_folder / path("weak.bin"), weak);
		if (image.empty()) image = Mat::zeros(depth.rows, depth.cols, CV_8UC3);
		camera.width = image.cols; camera.height = image.rows;
		images.emplace_back(RescaleColor(image, depth.cols, depth.rows, camera));
		cameras.emplace_back(camera);
		depths.emplace_back(depth);
		normals.emplace_back(normal);
		masks.emplace_back(Mat::zeros(depth.rows, depth.cols, CV_8UC1));
		RescaleMatToTargetSize<uint8_t>(weak, weak, depth.cols, depth.rows);
		weaks.emplace_back(weak);
	}
	std::vector<PointList> PointCloud;
	for (int i = 0; i < num_images; ++i) {
		std::cout << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		const auto& problem = problems[i];
		int ref_index = imageIdToindexMap[problem.ref_image_id];
		const int cols = depths[ref_index].cols, rows = depths[ref_index].rows;
		int num_ngb = (int)problem.src_image_ids.size();
		for (int r = 0; r < rows; ++r) {
			for (int c = 0; c < cols; ++c) {
				if (masks[ref_index].at<uint8_t>(r, c) == 1) continue;
				float ref_depth = depths[ref_index].at<float>(r, c);
				if (ref_depth <= 0.0) continue;
				const Vec3f ref_normal = normals[ref_index].at<Vec3f>(r, c);
				float3 PointX = Get3DPointonWorld(c, r, ref_depth, cameras[ref_index]);
				float3 consistent_Point = PointX;
				int num_consistent = 0;
				float dynamic_consistency = 0.0f;
				std::vector<int2> used_list(num_ngb, int2{-1, -1});
				for (int j = 0; j < num_ngb; ++j) {
					auto it = imageIdToindexMap.find(problem.src_image_ids[j]);
					if (it == imageIdToindexMap.end()) continue;
					int src_index = it->second;
					const int src_cols = depths[src_index].cols, src_rows = depths[src_index].rows;
					float2 point;
					float proj_depth;
					ProjectCamera(PointX, cameras[src_index], point, proj_depth);
					int src_r = int(point.y + 0.5f);
					int src_c = int(point.x + 0.5f);
					if (src_c >= 0 && src_c < src_cols && src_r >= 0 && src_r < src_rows) {
						if (masks[src_index].at<uint8_t>(src_r, src_c) == 1) continue;
						float src_depth = depths[src_index].at<float>(src_r, src_c);
						if (src_depth <= 0.0) continue;
						const Vec3f src_normal = normals[src_index].at<Vec3f>(src_r, src_c);
						float3 tmp_X = Get3DPointonWorld(src_c, src_r, src_depth, cameras[src_index]);
						float2 tmp_pt;
						ProjectCamera(tmp_X, cameras[ref_index], tmp_pt, proj_depth);
						float reproj_error = (float)std::sqrt(std::pow(c - tmp_pt.x, 2) + std::pow(r - tmp_pt.y, 2));
						float relative_depth_diff = std::fabs(proj_depth - ref_depth) / ref_depth;
						float angle = GetAngle(ref_normal, src_normal);
						if (reproj_error < 2.0f && relative_depth_diff < 0.01f && angle < 0.174533f) {
							used_list[j].x = src_c;
							used_list[j].y = src_r;
							float tmp_index = reproj_error + 200 * relative_depth_diff + angle * 10;
							dynamic_consistency += (float)std::exp(-tmp_index);
							num_consistent++;
						}
					}
				}
				float factor = (weaks[ref_index].at<uint8_t>(r, c) == WEAK ? 0.45f : 0.3f);
				if (num_consistent >= 1 && (dynamic_consistency > factor * num_consistent)) {
					PointList point3D;
					point3D.coord = consistent_Point;
					const uint8_t* px = images[ref_index].data + (size_t)r * images[ref_index].step + 3 * c;
					float col[3] = { (float)px[0], (float)px[1], (float)px[2] };
					for (int j = 0; j < num_ngb; ++j) {
						if (used_list[j].x == -1) continue;
						int src_index = imageIdToindexMap[problem.src_image_ids[j]];
						masks[src_index].at<uint8_t>(used_list[j].y, used_list[j].x) = 1;
						const uint8_t* q = images[src_index].data + (size_t)used_list[j].y * images[src_index].step + 3 * used_list[j].x;
						col[0] += q[0]; col[1] += q[1]; col[2] += q[2];
					}
					col[0] /= (num_consistent + 1); col[1] /= (num_consistent + 1); col[2] /= (num_consistent + 1);
					point3D.color = float3{col[0], col[1], col[2]};
					PointCloud.emplace_back(point3D);
				}
			}
		}
	}
	path ply_path = dense_folder / path("APD") / path("APD.ply");
	ExportPointCloud(ply_path, PointCloud);
	std::cout << "Fusion: " << PointCloud.size() << " points -> " << ply_path << std::endl;
}
