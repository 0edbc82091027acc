/*
 * oracle/ref_host/gaussian_model.h -- STAND-IN (ours) for the reference's include/gaussian_model.h, TEST INFRASTRUCTURE ONLY.
 *
 * The reference's header (include/gaussian_model.h:18-36) pulls in Sophus, Eigen, OpenCV (through tensor_utils.h) and
 * ORB-SLAM3 types, none of which exist in this image.  This file declares class GaussianModel with the members of
 * include/gaussian_model.h:59-193 -- same names, same types, same order of the data members -- minus createFromPcd (its
 * signature needs std::map<point3D_id_t, Point3D> with Eigen members) and saveSparsePointsPly (never called on the path);
 * applyScaledTransformation takes the stand-in Sophus::SE3f of sophus_standin.h.  The member FUNCTIONS are not restated anywhere: oracle/build_ref.py
 * extracts them verbatim, by name, from /root/reference/src/gaussian_model.cpp into a generated include file.
 *
 * Who includes it: the reference's own src/gaussian_renderer.cpp / src/gaussian_trainer.cpp / include/gaussian_renderer.h
 * (compiled verbatim; `#include "gaussian_model.h"` resolves here through the include tree oracle/build_ref.py lays out),
 * oracle/ref_densify.cpp and oracle/ref_host.cpp.
 *
 * REF_DEVICE: torch::kCUDA (= the HIP device of a ROCm LibTorch) unless the including translation unit maps the reference's
 * hard-coded kCUDA to the host for this GPU-less container (oracle/ref_host/emu_device.h, oracle/ref_densify.cpp).
 */
#pragma once

#include <memory>
#include <string>
#include <filesystem>
#include <fstream>
#include <algorithm>
#include <mutex>
#include <vector>

#include <torch/torch.h>
#ifndef REF_HOST_NO_CACHING_ALLOCATOR
#include <c10/cuda/CUDACachingAllocator.h>   /* as include/gaussian_model.h:25; on ROCm: host/include/compat/c10/cuda/ */
#endif

#include "third_party/simple-knn/spatial.h"  /* the reference's own declaration of distCUDA2 */
#include "third_party/tinyply/tinyply.h"
#include "types.h"
#include "operate_points.h"
#include "general_utils.h"
#include "sh_utils.h"
#include "gaussian_parameters.h"
#include "sophus_standin.h"                  /* ours: Sophus::SE3f / tensor_utils::EigenMatrix2TorchTensor for one signature */

#ifndef REF_DEVICE
#define REF_DEVICE torch::kCUDA
#endif

#define GAUSSIAN_MODEL_TENSORS_TO_VEC                        \
    this->Tensor_vec_xyz_ = {this->xyz_};                    \
    this->Tensor_vec_feature_dc_ = {this->features_dc_};     \
    this->Tensor_vec_feature_rest_ = {this->features_rest_}; \
    this->Tensor_vec_opacity_ = {this->opacity_};            \
    this->Tensor_vec_scaling_ = {this->scaling_};            \
    this->Tensor_vec_rotation_ = {this->rotation_};

class GaussianModel
{
public:
    GaussianModel() = default;   /* the reference's constructors take a degree / GaussianModelParams and only initialise members */

    torch::Tensor getScalingActivation();
    torch::Tensor getRotationActivation();
    torch::Tensor getXYZ();
    torch::Tensor getFeatures();
    torch::Tensor getOpacityActivation();
    torch::Tensor getCovarianceActivation(int scaling_modifier = 1);

    void oneUpShDegree();
    void setShDegree(const int sh);

    void increasePcd(std::vector<float> points, std::vector<float> colors, const int iteration);
    void increasePcd(torch::Tensor& new_point_cloud, torch::Tensor& new_colors, const int iteration);

    void applyScaledTransformation(const float s, const Sophus::SE3f T);   /* (the reference's default arguments need Eigen) */
    void scaledTransformationPostfix(
        torch::Tensor& new_xyz,
        torch::Tensor& new_scaling);

    void scaledTransformVisiblePointsOfKeyframe(
        torch::Tensor& point_not_transformed_flags,
        torch::Tensor& diff_pose,
        torch::Tensor& kf_world_view_transform,
        torch::Tensor& kf_full_proj_transform,
        const int kf_creation_iter,
        const int stable_num_iter_existence,
        int& num_transformed,
        const float scale = 1.0f);

    void trainingSetup(const GaussianOptimizationParams& training_args);
    float updateLearningRate(int step);
    void setPositionLearningRate(float position_lr);
    void setFeatureLearningRate(float feature_lr);
    void setOpacityLearningRate(float opacity_lr);
    void setScalingLearningRate(float scaling_lr);
    void setRotationLearningRate(float rot_lr);

    void resetOpacity();
    torch::Tensor replaceTensorToOptimizer(torch::Tensor& t, int tensor_idx);

    void prunePoints(torch::Tensor& mask);

    void densificationPostfix(
        torch::Tensor& new_xyz,
        torch::Tensor& new_features_dc,
        torch::Tensor& new_features_rest,
        torch::Tensor& new_opacities,
        torch::Tensor& new_scaling,
        torch::Tensor& new_rotation,
        torch::Tensor& new_exist_since_iter);

    void densifyAndSplit(
        torch::Tensor& grads,
        float grad_threshold,
        float scene_extent,
        int N = 2);

    void densifyAndClone(
        torch::Tensor& grads,
        float grad_threshold,
        float scene_extent);

    void densifyAndPrune(
        float max_grad,
        float min_opacity,
        float extent,
        int max_screen_size);

    void addDensificationStats(
        torch::Tensor& viewspace_point_tensor,
        torch::Tensor& update_filter);

    void loadPly(std::filesystem::path ply_path);
    void savePly(std::filesystem::path result_path);

    float percentDense();
    void setPercentDense(const float percent_dense);

protected:
    float exponLrFunc(int step);

public:
    torch::DeviceType device_type_ = REF_DEVICE;

    int active_sh_degree_ = 0;
    int max_sh_degree_ = 3;

    torch::Tensor xyz_;
    torch::Tensor features_dc_;
    torch::Tensor features_rest_;
    torch::Tensor scaling_;
    torch::Tensor rotation_;
    torch::Tensor opacity_;
    torch::Tensor max_radii2D_;
    torch::Tensor xyz_gradient_accum_;
    torch::Tensor denom_;
    torch::Tensor exist_since_iter_;

    std::vector<torch::Tensor> Tensor_vec_xyz_,
                               Tensor_vec_feature_dc_,
                               Tensor_vec_feature_rest_,
                               Tensor_vec_opacity_,
                               Tensor_vec_scaling_ ,
                               Tensor_vec_rotation_;

    std::shared_ptr<torch::optim::Adam> optimizer_;
    float percent_dense_ = 0.01f;
    float spatial_lr_scale_ = 1.0f;

    torch::Tensor sparse_points_xyz_ = torch::empty({0, 3});
    torch::Tensor sparse_points_color_ = torch::empty({0, 3});

protected:
    float lr_init_ = 0.f;
    float lr_final_ = 0.f;
    int lr_delay_steps_ = 0;
    float lr_delay_mult_ = 1.f;
    int max_steps_ = 1000000;

    std::mutex mutex_settings_;
};
