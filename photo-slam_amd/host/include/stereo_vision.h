// stereo_vision.h -- reprojectDepthPinhole / monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints
// with the reference's declarations (include/stereo_vision.h; implementation src/stereo_vision.cu:138-215).
#pragma once
#include <torch/torch.h>

#include <tuple>
#include <vector>

torch::Tensor reprojectDepthPinhole(torch::Tensor& depth, torch::Tensor& mask, std::vector<float>& intr, int width);

// returns (pt3D, colors of pt3D) of the keypoints that end up with a positive depth
std::tuple<torch::Tensor, torch::Tensor> monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(
    torch::Tensor& kps_pixel, torch::Tensor& kps_has3D, torch::Tensor& kps_point_local, torch::Tensor& colors,
    float max_pixel_dist, std::vector<float>& intr, int width);
