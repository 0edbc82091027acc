// operate_points.h -- transformPoints / scaleAndTransformThenMarkVisiblePoints with the reference's
// declarations (include/operate_points.h of Photo-SLAM; implementation src/operate_points.cu:73-143).
#pragma once
#include <torch/torch.h>

void transformPoints(torch::Tensor& points, torch::Tensor& transformmatrix);

void scaleAndTransformThenMarkVisiblePoints(torch::Tensor& points, torch::Tensor& rots,
                                            torch::Tensor& point_not_transformed_mask, torch::Tensor& point_unstable_mask,
                                            torch::Tensor& transformmatrix, torch::Tensor& viewmatrix,
                                            torch::Tensor& projmatrix, int& num_transformed, const float scale = 1.0f);
