// gaussian_renderer.h -- GaussianRenderer::render (include/gaussian_renderer.h:29-42).
//
// Inside the Photo-SLAM tree this header is used with the reference's own gaussian_model.h /
// gaussian_keyframe.h / gaussian_parameters.h (define PHOTOSLAM_TREE).  Stand-alone (this repo:
// no OpenCV / Eigen / ORB-SLAM3 available) render() is a template over the model and keyframe
// types and touches exactly the members the reference implementation touches
// (src/gaussian_renderer.cpp:41-148): FoVx_, FoVy_, world_view_transform_, full_proj_transform_,
// camera_center_ ; getXYZ(), getOpacityActivation(), getScalingActivation(),
// getRotationActivation(), getCovarianceActivation(), getFeatures(), active_sh_degree_.
#pragma once
#include <torch/torch.h>

#include <cmath>
#include <cstdlib>
#include <memory>
#include <tuple>

#include "gaussian_rasterizer.h"
#include "sh_utils.h"

#ifdef PHOTOSLAM_TREE
#include "gaussian_keyframe.h"
#include "gaussian_model.h"
#include "gaussian_parameters.h"
#elif !defined(GSR_HAVE_PIPELINE_PARAMS)
#define GSR_HAVE_PIPELINE_PARAMS
struct GaussianPipelineParams {   // include/gaussian_parameters.h:41-49
	bool convert_SHs_ = false;
	bool compute_cov3D_ = false;
};
#endif

class GaussianRenderer {
public:
	// returns (render, viewspace_points, visibility_filter, radii)
	template <class Keyframe, class Model>
	static std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> render(
	    std::shared_ptr<Keyframe> viewpoint_camera, int image_height, int image_width, std::shared_ptr<Model> pc,
	    GaussianPipelineParams& pipe, torch::Tensor& bg_color, torch::Tensor& override_color,
	    float scaling_modifier = 1.0f, bool use_override_color = false, bool fuse_activations = false,
	    torch::Tensor sh_grad_view = torch::Tensor() /* extension: GaussianRasterizationExtensions::sh_grad_view_ */,
	    ShAdamStep sh_adam = ShAdamStep() /* extension: GaussianRasterizationExtensions::sh_adam_ */,
	    std::vector<torch::Tensor> view_stats = {} /* extension: GaussianRasterizationExtensions::view_stats_ */,
	    GeomAdamStep geom_adam = GeomAdamStep() /* extension: GaussianRasterizationExtensions::geom_adam_ */,
	    bool cull_empty_tiles = false /* extension: GaussianRasterizationExtensions::cull_empty_tiles_ */,
	    RasterWorkspace* workspace = nullptr /* extension: GaussianRasterizationExtensions::workspace_ */)
	{
		// fuse_activations (extension): hand the raw opacity_/scaling_/rotation_ leaves to the rasterizer, which applies
		// sigmoid / exp / normalize and their chain rule in-kernel (include/gsr.h raw_params)
		// dummy input whose gradient is dL/dmean2D (the densification statistic)
		// (with geom_adam.training_outputs_only nobody reads its gradient and the rasterizer never reads its values: the 12 P
		// bytes are then not even zero-filled)
		auto screenspace_points = geom_adam.training_outputs_only
		                              ? torch::empty_like(pc->getXYZ(), torch::TensorOptions().requires_grad(true))
		                              : torch::zeros_like(pc->getXYZ(), torch::TensorOptions().requires_grad(true));
		screenspace_points.retain_grad();

		const float tanfovx = std::tan(viewpoint_camera->FoVx_ * 0.5f);
		const float tanfovy = std::tan(viewpoint_camera->FoVy_ * 0.5f);
		GaussianRasterizationSettings raster_settings(image_height, image_width, tanfovx, tanfovy, bg_color,
		                                              scaling_modifier, viewpoint_camera->world_view_transform_,
		                                              viewpoint_camera->full_proj_transform_, pc->active_sh_degree_,
		                                              viewpoint_camera->camera_center_, false);
		// SH evaluated in torch (convert_SHs_) or colours given: the rasterizer sees no SH tensor, so the SH extensions are off
		const bool sh_in_rasterizer = !use_override_color && !pipe.convert_SHs_;
		GaussianRasterizationExtensions ext;
		ext.raw_params_ = (fuse_activations && !pipe.compute_cov3D_) ? 7 : 0;
		if (sh_in_rasterizer) ext.sh_grad_view_ = sh_grad_view;
		if (sh_in_rasterizer) ext.sh_adam_ = sh_adam;
		ext.view_stats_ = view_stats;
		ext.workspace_ = workspace;
		// (the same image and gradients either way; off by default: measured a wash, DESIGN.md section 10.  The caller's
		// argument decides; the environment variable GSR_CULL_EMPTY_TILES=0/1, when set, overrides it -- an A/B handle)
		static const int cull_env = [] { const char* e = std::getenv("GSR_CULL_EMPTY_TILES"); return (e && *e) ? (e[0] == '1' ? 1 : 0) : -1; }();
		ext.cull_empty_tiles_ = cull_env >= 0 ? cull_env != 0 : cull_empty_tiles;
		// the fused geometry step needs the raw leaves in the rasterizer (it steps opacity_ / scaling_ / rotation_ themselves)
		if (ext.raw_params_ == 7 && !pipe.compute_cov3D_) ext.geom_adam_ = geom_adam;
		GaussianRasterizerEx rasterizer(raster_settings, ext);

		auto means3D = pc->getXYZ();
		auto opacity = ext.raw_params_ ? pc->opacity_ : pc->getOpacityActivation();
		bool has_scales = false, has_rotations = false, has_cov3D_precomp = false;
		torch::Tensor scales, rotations, cov3D_precomp;
		if (pipe.compute_cov3D_) {
			cov3D_precomp = pc->getCovarianceActivation();
			has_cov3D_precomp = true;
		} else {
			scales = ext.raw_params_ ? pc->scaling_ : pc->getScalingActivation();
			rotations = ext.raw_params_ ? pc->rotation_ : pc->getRotationActivation();
			has_scales = has_rotations = true;
		}
		bool has_shs = false, has_color_precomp = false;
		torch::Tensor shs, colors_precomp;
		if (use_override_color) {
			colors_precomp = override_color;
			has_color_precomp = true;
		} else if (pipe.convert_SHs_) {
			// src/gaussian_renderer.cpp:106-113: SH -> RGB in torch, handed to the rasterizer as colors_precomp
			const int max_coeffs = (pc->max_sh_degree_ + 1) * (pc->max_sh_degree_ + 1);
			auto shs_view = pc->getFeatures().transpose(1, 2).reshape({-1, 3, max_coeffs});
			auto dir_pp = pc->getXYZ() - viewpoint_camera->camera_center_.reshape({1, 3});
			auto dir_pp_normalized = dir_pp / torch::norm(dir_pp, 2, {1}, /*keepdim=*/true);
			auto sh2rgb = sh_utils::eval_sh(pc->active_sh_degree_, shs_view, dir_pp_normalized);
			colors_precomp = torch::clamp_min(sh2rgb + 0.5, 0.0);
			has_color_precomp = true;
		} else {
			shs = pc->getFeatures();
			has_shs = true;
		}
		auto result = rasterizer.forward(means3D, screenspace_points, opacity, has_shs, has_color_precomp, has_scales,
		                                 has_rotations, has_cov3D_precomp, shs, colors_precomp, scales, rotations,
		                                 cov3D_precomp);
		auto rendered_image = std::get<0>(result);
		auto radii = std::get<1>(result);
		// (visibility_filter = radii > 0 is one more launch: a caller that fused everything that consumes it -- the statistics,
		// geom_adam.training_outputs_only -- gets an undefined tensor and derives it from radii if it ever wants it)
		return std::make_tuple(rendered_image, screenspace_points,
		                       geom_adam.training_outputs_only ? torch::Tensor() : (radii > 0), radii);
	}
};
