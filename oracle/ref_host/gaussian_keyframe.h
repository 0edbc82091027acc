/*
 * oracle/ref_host/gaussian_keyframe.h -- STAND-IN (ours) for the reference's include/gaussian_keyframe.h, TEST INFRASTRUCTURE ONLY.
 *
 * The reference's header needs OpenCV, Eigen and Sophus (include/gaussian_keyframe.h:24-35).  The hot path reads nine of the
 * keyframe's data members (src/gaussian_renderer.cpp:51-64, src/gaussian_trainer.cpp:66-82): they are declared here with the
 * reference's names and types; the harness (oracle/ref_host.cpp) fills the transform tensors the way
 * GaussianKeyframe::computeTransformTensors does (src/gaussian_keyframe.cpp:119-152).
 */
#pragma once

#include <cstddef>
#include <memory>

#include <torch/torch.h>

class GaussianKeyframe
{
public:
    GaussianKeyframe() {}

    GaussianKeyframe(std::size_t fid, int creation_iter = 0)
        : fid_(fid), creation_iter_(creation_iter) {}

public:
    std::size_t fid_ = 0;
    int creation_iter_ = 0;

    torch::Tensor original_image_; ///< image
    int image_width_ = 0;          ///< image
    int image_height_ = 0;         ///< image

    float FoVx_ = 0.f; ///< intrinsics
    float FoVy_ = 0.f; ///< intrinsics

    float zfar_ = 100.0f;
    float znear_ = 0.01f;

    torch::Tensor world_view_transform_;    ///< transform tensors
    torch::Tensor projection_matrix_;       ///< transform tensors
    torch::Tensor full_proj_transform_;     ///< transform tensors
    torch::Tensor camera_center_;           ///< transform tensors
};
