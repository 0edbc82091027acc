/*
 * oracle/ref_host/gaussian_trainer.h -- STAND-IN (ours) for the reference's include/gaussian_trainer.h, TEST INFRASTRUCTURE ONLY.
 *
 * The reference's header includes <opencv2/opencv.hpp> and "ORB-SLAM3/include/System.h" (include/gaussian_trainer.h:31-33),
 * which src/gaussian_trainer.cpp does not use.  The class declaration below is the reference's (:41-72), so that
 * src/gaussian_trainer.cpp compiles VERBATIM against it.
 */
#pragma once

#include <torch/torch.h>

#include <iomanip>
#include <iostream>
#include <random>
#include <chrono>
#include <memory>
#include <thread>
#include <mutex>
#include <vector>
#include <unordered_map>
#include <functional>

#include "loss_utils.h"
#include "gaussian_parameters.h"
#include "gaussian_model.h"
#include "gaussian_scene.h"
#include "gaussian_renderer.h"

class GaussianTrainer
{
public:
    GaussianTrainer();

    static void trainingOnce(
        std::shared_ptr<GaussianScene> scene,
        std::shared_ptr<GaussianModel> gaussians,
        GaussianModelParams& dataset,
        GaussianOptimizationParams& opt,
        GaussianPipelineParams& pipe,
        torch::DeviceType device_type = torch::kCUDA,
        std::vector<int> testing_iterations = {},
        std::vector<int> saving_iterations = {},
        std::vector<int> checkpoint_iterations = {}/*, checkpoint*/);

    static void trainingReport(
        int iteration,
        int num_iterations,
        torch::Tensor& Ll1,
        torch::Tensor& loss,
        float ema_loss_for_log,
        std::function<torch::Tensor(torch::Tensor&, torch::Tensor&)> l1_loss,
        int64_t elapsed_time,
        GaussianModel& gaussians,
        GaussianScene& scene,
        GaussianPipelineParams& pipe,
        torch::Tensor& background);

};
