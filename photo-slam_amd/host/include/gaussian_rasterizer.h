// gaussian_rasterizer.h -- GaussianRasterizationSettings / GaussianRasterizerFunction / rasterizeGaussians /
// GaussianRasterizer declared EXACTLY as the reference declares them (include/gaussian_rasterizer.h:25-127): same members
// in the same order (= the same object layout), same signatures (= the same mangled symbols), so that
// GaussianRenderer::render and the mapper call sites (src/gaussian_renderer.cpp:51-66,129-148) compile unchanged AND an
// object compiled against the reference's header links and runs against libphotoslam_host.so
// (tests/test_reference_link.py).  This repository's extensions live in the separate *Ex types below.
#pragma once
#include <torch/torch.h>

#include <tuple>
#include <vector>

#include "rasterize_points.h"

struct GaussianRasterizationSettings {
	GaussianRasterizationSettings(int image_height, int image_width, float tanfovx, float tanfovy, torch::Tensor& bg,
	                              float scale_modifier, torch::Tensor& viewmatrix, torch::Tensor& projmatrix,
	                              int sh_degree, torch::Tensor& campos, bool prefiltered)
	    : image_height_(image_height), image_width_(image_width), tanfovx_(tanfovx), tanfovy_(tanfovy), bg_(bg),
	      scale_modifier_(scale_modifier), viewmatrix_(viewmatrix), projmatrix_(projmatrix), sh_degree_(sh_degree),
	      campos_(campos), prefiltered_(prefiltered)
	{
	}
	int image_height_;
	int image_width_;
	float tanfovx_;
	float tanfovy_;
	torch::Tensor bg_;
	float scale_modifier_;
	torch::Tensor viewmatrix_;
	torch::Tensor projmatrix_;
	int sh_degree_;
	torch::Tensor campos_;
	bool prefiltered_;
};

class GaussianRasterizerFunction : public torch::autograd::Function<GaussianRasterizerFunction> {
public:
	static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D,
	                                            torch::Tensor means2D, torch::Tensor sh, torch::Tensor colors_precomp,
	                                            torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
	                                            torch::Tensor cov3Ds_precomp,
	                                            GaussianRasterizationSettings raster_settings);
	static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
	                                             torch::autograd::tensor_list grad_out_color);
};

inline torch::autograd::tensor_list rasterizeGaussians(torch::Tensor& means3D, torch::Tensor& means2D, torch::Tensor& sh,
                                                       torch::Tensor& colors_precomp, torch::Tensor& opacities,
                                                       torch::Tensor& scales, torch::Tensor& rotations,
                                                       torch::Tensor& cov3Ds_precomp,
                                                       GaussianRasterizationSettings& raster_settings)
{
	return GaussianRasterizerFunction::apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
	                                         cov3Ds_precomp, raster_settings);
}

class GaussianRasterizer : public torch::nn::Module {
public:
	GaussianRasterizer(GaussianRasterizationSettings& raster_settings) : raster_settings_(raster_settings) {}

	torch::Tensor markVisibleGaussians(torch::Tensor& positions);

	std::tuple<torch::Tensor, torch::Tensor> forward(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities,
	                                                 bool has_shs, bool has_colors_precomp, bool has_scales,
	                                                 bool has_rotations, bool has_cov3D_precomp, torch::Tensor shs,
	                                                 torch::Tensor colors_precomp, torch::Tensor scales,
	                                                 torch::Tensor rotations, torch::Tensor cov3D_precomp);

public:
	GaussianRasterizationSettings raster_settings_;
};

// ---- extensions of this repository (none of them alters the types above) ------------------------------------------------
struct GaussianRasterizationExtensions {
	int raw_params_ = 0;   // GSR_RAW_* mask (activations fused into the rasterizer), see include/gsr.h
	// a [P,3] tensor that receives the clamp-masked colour gradient in backward; sh then gets no gradient from autograd --
	// the view-factored exchange of the data-parallel step rebuilds it (shGradFromViews)
	torch::Tensor sh_grad_view_;
	// optimizer-in-backward for the SH tensor (rasterize_points.h): set exp_avg to enable
	ShAdamStep sh_adam_;
	// {xyz_gradient_accum, denom, max_radii2D} -- backward adds this view's densification statistics itself
	std::vector<torch::Tensor> view_stats_;
	// optimizer-in-backward for xyz / opacity / scaling / rotation (rasterize_points.h): fill param to enable; those four then
	// get no gradient from autograd
	GeomAdamStep geom_adam_;
	// GSR_CULL_EMPTY_TILES (include/gsr.h): instances of tiles in which no pixel can blend the Gaussian are dropped in front of
	// the tile sort -- the same image and the same gradients from shorter internal lists
	bool cull_empty_tiles_ = false;
	// persistent scratch buffers of a caller that renders iteration after iteration (rasterize_points.h: RasterWorkspace; the
	// caller owns it and keeps it alive until the backward pass has run); nullptr = fresh buffers per call, as the reference
	RasterWorkspace* workspace_ = nullptr;
};

class GaussianRasterizerFunctionEx : public torch::autograd::Function<GaussianRasterizerFunctionEx> {
public:
	static torch::autograd::tensor_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D,
	                                            torch::Tensor means2D, torch::Tensor sh, torch::Tensor colors_precomp,
	                                            torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
	                                            torch::Tensor cov3Ds_precomp, GaussianRasterizationSettings raster_settings,
	                                            GaussianRasterizationExtensions extensions);
	static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx,
	                                             torch::autograd::tensor_list grad_out_color);
};

// GaussianRasterizer with the extensions: same forward() contract and exception texts
class GaussianRasterizerEx : public GaussianRasterizer {
public:
	GaussianRasterizerEx(GaussianRasterizationSettings& raster_settings, const GaussianRasterizationExtensions& extensions)
	    : GaussianRasterizer(raster_settings), extensions_(extensions)
	{
	}
	std::tuple<torch::Tensor, torch::Tensor> forward(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities,
	                                                 bool has_shs, bool has_colors_precomp, bool has_scales,
	                                                 bool has_rotations, bool has_cov3D_precomp, torch::Tensor shs,
	                                                 torch::Tensor colors_precomp, torch::Tensor scales,
	                                                 torch::Tensor rotations, torch::Tensor cov3D_precomp);
	GaussianRasterizationExtensions extensions_;
};
