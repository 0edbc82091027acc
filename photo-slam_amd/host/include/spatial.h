// spatial.h -- distCUDA2 (third_party/simple-knn/spatial.h:14): mean squared distance to the three
// nearest neighbours of every point, used to initialise Gaussian scales
// (src/gaussian_model.cpp:155,238,325).  Implemented by gsr_knn_mean_dist2 (include/gsr.h).
#pragma once
#include <torch/torch.h>

torch::Tensor distCUDA2(const torch::Tensor& points);
