// operate_points.cpp -- host side of Photo-SLAM's point-cloud operators on top of the C-ABI (csrc/points.hip).
#include "operate_points.h"

#include <stdexcept>
#include <string>

#include "../../../include/gsr.h"
#include "rasterize_points.h"
#include "stereo_vision.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif

namespace {
void* stream_of(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}
void check(int status, const char* where)
{
	if (status != GSR_OK) throw std::runtime_error(std::string(where) + ": " + gsr_strerror(status));
}
}  // namespace

void transformPoints(torch::Tensor& points, torch::Tensor& transformmatrix)
{
	if (points.ndimension() != 2 || points.size(1) != 3) {
		AT_ERROR("points must have dimensions (num_points, 3)");
	}
	const int P = static_cast<int>(points.size(0));
	if (P == 0) return;
	auto pts = points.contiguous(), m = transformmatrix.contiguous();
	auto out = torch::zeros_like(pts);
	check(gsr_transform_points(P, pts.data_ptr<float>(), m.data_ptr<float>(), out.data_ptr<float>(), stream_of(pts)),
	      "transformPoints");
	points = out;
}

void scaleAndTransformThenMarkVisiblePoints(torch::Tensor& points, torch::Tensor& rots,
                                            torch::Tensor& point_not_transformed_mask, torch::Tensor& point_unstable_mask,
                                            torch::Tensor& transformmatrix, torch::Tensor& viewmatrix,
                                            torch::Tensor& projmatrix, int& num_transformed, const float scale)
{
	if (points.ndimension() != 2 || points.size(1) != 3) {
		AT_ERROR("points must have dimensions (num_points, 3)");
	}
	torch::Tensor present = markVisible(points, viewmatrix, projmatrix);
	const auto num_points = present.size(0);
	if (point_not_transformed_mask.size(0) != num_points || point_unstable_mask.size(0) != num_points) {
		AT_ERROR("points_mask must have dimensions (num_points)");
	}
	torch::Tensor final_mask = torch::logical_and(torch::logical_and(point_not_transformed_mask, point_unstable_mask), present);
	num_transformed += final_mask.sum().item<int>();
	const int P = static_cast<int>(points.size(0));
	if (P == 0) return;
	auto pts = points.contiguous(), r = rots.contiguous(), m = transformmatrix.contiguous();
	auto tp = torch::zeros_like(pts), tr = torch::zeros_like(r);
	auto mk = final_mask.to(torch::kUInt8).contiguous();
	// reference_rot_layout = 1: bit-compatible with insert_rot_to_rots as shipped (cuda_rasterizer/operate_points.h:175-178)
	check(gsr_scale_transform_points(P, scale, pts.data_ptr<float>(), r.data_ptr<float>(), m.data_ptr<float>(),
	                                 mk.data_ptr<uint8_t>(), tp.data_ptr<float>(), tr.data_ptr<float>(), 1, stream_of(pts)),
	      "scaleAndTransformThenMarkVisiblePoints");
	points.index_put_({final_mask}, tp.index({final_mask}));
	rots.index_put_({final_mask}, tr.index({final_mask}));
	point_not_transformed_mask.index_put_({final_mask}, false);
}

torch::Tensor reprojectDepthPinhole(torch::Tensor& depth, torch::Tensor& mask, std::vector<float>& intr, int width)
{
	if (depth.ndimension() != 1) {
		AT_ERROR("points must have dimensions (num_points)");
	}
	const int P = static_cast<int>(depth.size(0));
	torch::Tensor points;
	if (P != 0) {
		points = torch::zeros({P, 3}, depth.options());
		auto d = depth.contiguous();
		auto mk = mask.to(torch::kUInt8).contiguous();
		check(gsr_reproject_depth_pinhole(P, width, intr[0], intr[1], intr[2], intr[3], d.data_ptr<float>(),
		                                  mk.data_ptr<uint8_t>(), points.data_ptr<float>(), stream_of(d)),
		      "reprojectDepthPinhole");
	}
	return points;
}

std::tuple<torch::Tensor, torch::Tensor> monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(
    torch::Tensor& kps_pixel, torch::Tensor& kps_has3D, torch::Tensor& kps_point_local, torch::Tensor& colors,
    float max_pixel_dist, std::vector<float>& intr, int width)
{
	if (kps_pixel.ndimension() != 2 || kps_pixel.size(1) != 2) AT_ERROR("kps_pixel must have dimensions (num_points, 2)");
	if (kps_has3D.ndimension() != 1) AT_ERROR("kps_has3D must have dimensions (num_points)");
	if (kps_point_local.ndimension() != 2 || kps_point_local.size(1) != 3)
		AT_ERROR("kps_point_local must have dimensions (num_points, 3)");
	const int N = static_cast<int>(kps_pixel.size(0));
	torch::Tensor result_pt, result_color;
	if (N != 0) {
		auto px = kps_pixel.contiguous(), p3 = kps_point_local.contiguous(), col = colors.contiguous();
		auto has = kps_has3D.to(torch::kUInt8).contiguous();
		result_pt = torch::zeros_like(p3);
		result_color = torch::zeros_like(p3);
		check(gsr_neighborhood_depth_pinhole(N, width, intr[0], intr[1], intr[2], intr[3], max_pixel_dist, px.data_ptr<float>(),
		                                     has.data_ptr<uint8_t>(), p3.data_ptr<float>(), col.data_ptr<float>(),
		                                     result_pt.data_ptr<float>(), result_color.data_ptr<float>(), stream_of(px)),
		      "monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints");
		auto valid = result_pt.index({torch::indexing::Slice(), 2}) > 0.0f;
		result_pt = result_pt.index({valid});
		result_color = result_color.index({valid});
	}
	return std::make_tuple(result_pt, result_color);
}
