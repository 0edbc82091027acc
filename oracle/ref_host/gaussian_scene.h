/*
 * oracle/ref_host/gaussian_scene.h -- STAND-IN (ours) for the reference's include/gaussian_scene.h, TEST INFRASTRUCTURE ONLY.
 *
 * GaussianTrainer::trainingOnce reads `scene->keyframes()` and `scene->cameras_extent_` (src/gaussian_trainer.cpp:58,121);
 * the reference's class also holds COLMAP cameras / cached 3-D points behind Eigen and Sophus types
 * (include/gaussian_scene.h:27-84).
 */
#pragma once

#include <map>
#include <memory>

#include "gaussian_parameters.h"
#include "gaussian_model.h"
#include "gaussian_keyframe.h"

class GaussianScene
{
public:
    GaussianScene() {}

    std::map<std::size_t, std::shared_ptr<GaussianKeyframe>>& keyframes() { return keyframes_; }   /* src/gaussian_scene.cpp:96-99 */

public:
    float cameras_extent_ = 0.f; ///< scene_info.nerf_normalization["radius"]

    std::map<std::size_t, std::shared_ptr<GaussianKeyframe>> keyframes_;
};
