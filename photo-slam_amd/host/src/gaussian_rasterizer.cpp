// gaussian_rasterizer.cpp -- autograd glue, counterpart of src/gaussian_rasterizer.cpp:18-234.
#include "gaussian_rasterizer.h"

#include <stdexcept>

torch::Tensor GaussianRasterizer::markVisibleGaussians(torch::Tensor& positions)
{
	torch::NoGradGuard no_grad;
	return markVisible(positions, raster_settings_.viewmatrix_, raster_settings_.projmatrix_);
}

namespace {

// shared by GaussianRasterizerFunction (the reference's contract: no extensions) and GaussianRasterizerFunctionEx
torch::autograd::tensor_list forward_impl(torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor sh,
                                          torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales,
                                          torch::Tensor rotations, torch::Tensor cov3Ds_precomp,
                                          const GaussianRasterizationSettings& s, const GaussianRasterizationExtensions& e)
{
	auto r = RasterizeGaussiansCUDA(s.bg_, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier_,
	                                cov3Ds_precomp, s.viewmatrix_, s.projmatrix_, s.tanfovx_, s.tanfovy_,
	                                s.image_height_, s.image_width_, sh, s.sh_degree_, s.campos_, s.prefiltered_,
	                                e.raw_params_ | (e.cull_empty_tiles_ ? 8 /* GSR_CULL_EMPTY_TILES, include/gsr.h */ : 0),
	                                e.sh_adam_ /* lazy mode: visible rows are brought up to date first */, e.workspace_);
	// (no zero tensor for the unused gradient of `radii`: autograd would otherwise fill P ints per backward)
	ctx->set_materialize_grads(false);
	ctx->saved_data["num_rendered"] = std::get<0>(r);
	ctx->saved_data["scale_modifier"] = static_cast<double>(s.scale_modifier_);
	ctx->saved_data["tanfovx"] = static_cast<double>(s.tanfovx_);
	ctx->saved_data["tanfovy"] = static_cast<double>(s.tanfovy_);
	ctx->saved_data["sh_degree"] = s.sh_degree_;
	ctx->saved_data["raw_params"] = e.raw_params_;
	if (e.sh_grad_view_.defined()) ctx->saved_data["sh_grad_view"] = e.sh_grad_view_;
	if (e.sh_adam_.color_view_ready_stream)
		ctx->saved_data["color_view_ready_stream"] = static_cast<int64_t>(reinterpret_cast<intptr_t>(e.sh_adam_.color_view_ready_stream));
	if (e.sh_adam_.packed_view.defined()) {
		ctx->saved_data["packed_view"] = e.sh_adam_.packed_view;
		ctx->saved_data["packed_capacity"] = e.sh_adam_.packed_capacity;
	}
	if (!e.view_stats_.empty()) ctx->saved_data["view_stats"] = e.view_stats_;
	if (e.sh_adam_.exp_avg.defined()) {
		ctx->saved_data["sh_adam_m"] = e.sh_adam_.exp_avg;
		ctx->saved_data["sh_adam_v"] = e.sh_adam_.exp_avg_sq;
		ctx->saved_data["sh_adam_h"] = std::vector<double>{e.sh_adam_.lr, e.sh_adam_.lr_tail, e.sh_adam_.beta1, e.sh_adam_.beta2,
		                                                   e.sh_adam_.eps, static_cast<double>(e.sh_adam_.step)};
		if (e.sh_adam_.row_step.defined()) {   // lazy mode
			ctx->saved_data["sh_adam_row_step"] = e.sh_adam_.row_step;
			ctx->saved_data["sh_adam_window"] = e.sh_adam_.window;
			ctx->saved_data["sh_adam_lr_past"] = e.sh_adam_.lr_past;
			ctx->saved_data["sh_adam_lr_tail_past"] = e.sh_adam_.lr_tail_past;
		}
	}
	if (!e.geom_adam_.param.empty() || e.geom_adam_.training_outputs_only) {
		const auto& ga = e.geom_adam_;
		ctx->saved_data["geom_adam_p"] = ga.param;
		ctx->saved_data["geom_adam_m"] = ga.exp_avg;
		ctx->saved_data["geom_adam_v"] = ga.exp_avg_sq;
		ctx->saved_data["geom_adam_lr"] = ga.lr;
		ctx->saved_data["geom_adam_step"] = ga.step;
		ctx->saved_data["geom_adam_h"] = std::vector<double>{ga.beta1, ga.beta2, ga.eps, ga.training_outputs_only ? 1.0 : 0.0};
	}
	auto color = std::get<1>(r);
	auto radii = std::get<2>(r);
	// same 14 tensors, same order as the reference (src/gaussian_rasterizer.cpp:87-100)
	ctx->save_for_backward({s.bg_, s.viewmatrix_, s.projmatrix_, s.campos_, colors_precomp, means3D, scales, rotations,
	                        cov3Ds_precomp, radii, sh, std::get<3>(r), std::get<4>(r), std::get<5>(r)});
	ctx->mark_non_differentiable({radii});
	return {color, radii};
}

// n_extra: undefined gradients for the trailing non-tensor inputs (raster_settings [, extensions])
torch::autograd::tensor_list backward_impl(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list grad_outputs,
                                           int n_extra)
{
	if (!grad_outputs[0].defined()) {   // the image took no part in the loss (set_materialize_grads(false)): no gradients
		torch::autograd::tensor_list none(static_cast<size_t>(8 + n_extra));
		return none;
	}
	const int num_rendered = static_cast<int>(ctx->saved_data["num_rendered"].toInt());
	const float scale_modifier = static_cast<float>(ctx->saved_data["scale_modifier"].toDouble());
	const float tanfovx = static_cast<float>(ctx->saved_data["tanfovx"].toDouble());
	const float tanfovy = static_cast<float>(ctx->saved_data["tanfovy"].toDouble());
	const int sh_degree = static_cast<int>(ctx->saved_data["sh_degree"].toInt());
	const int raw_params = static_cast<int>(ctx->saved_data["raw_params"].toInt());
	torch::Tensor sh_grad_view;
	if (ctx->saved_data.count("sh_grad_view")) sh_grad_view = ctx->saved_data["sh_grad_view"].toTensor();
	ShAdamStep sh_adam;
	if (ctx->saved_data.count("sh_adam_m")) {
		sh_adam.exp_avg = ctx->saved_data["sh_adam_m"].toTensor();
		sh_adam.exp_avg_sq = ctx->saved_data["sh_adam_v"].toTensor();
		const auto h = ctx->saved_data["sh_adam_h"].toDoubleVector();
		sh_adam.lr = h[0]; sh_adam.lr_tail = h[1];
		sh_adam.beta1 = h[2]; sh_adam.beta2 = h[3];
		sh_adam.eps = h[4]; sh_adam.step = static_cast<int>(h[5]);
		if (ctx->saved_data.count("sh_adam_row_step")) {
			sh_adam.row_step = ctx->saved_data["sh_adam_row_step"].toTensor();
			sh_adam.window = static_cast<int>(ctx->saved_data["sh_adam_window"].toInt());
			sh_adam.lr_past = ctx->saved_data["sh_adam_lr_past"].toDoubleVector();
			sh_adam.lr_tail_past = ctx->saved_data["sh_adam_lr_tail_past"].toDoubleVector();
		}
	}
	std::vector<torch::Tensor> view_stats;
	if (ctx->saved_data.count("view_stats")) view_stats = ctx->saved_data["view_stats"].toTensorVector();
	GeomAdamStep geom_adam;
	if (ctx->saved_data.count("geom_adam_h")) {
		geom_adam.param = ctx->saved_data["geom_adam_p"].toTensorVector();
		geom_adam.exp_avg = ctx->saved_data["geom_adam_m"].toTensorVector();
		geom_adam.exp_avg_sq = ctx->saved_data["geom_adam_v"].toTensorVector();
		geom_adam.lr = ctx->saved_data["geom_adam_lr"].toDoubleVector();
		geom_adam.step = ctx->saved_data["geom_adam_step"].toIntVector();
		const auto h = ctx->saved_data["geom_adam_h"].toDoubleVector();
		geom_adam.beta1 = h[0]; geom_adam.beta2 = h[1]; geom_adam.eps = h[2];
		geom_adam.training_outputs_only = h[3] != 0.0;
	}
	ShAdamStep bwd_adam = (sh_grad_view.defined() && !sh_adam.row_step.defined()) ? ShAdamStep() : sh_adam;
	if (ctx->saved_data.count("color_view_ready_stream"))
		bwd_adam.color_view_ready_stream = reinterpret_cast<void*>(static_cast<intptr_t>(ctx->saved_data["color_view_ready_stream"].toInt()));
	if (ctx->saved_data.count("packed_view")) {
		bwd_adam.packed_view = ctx->saved_data["packed_view"].toTensor();
		bwd_adam.packed_capacity = ctx->saved_data["packed_capacity"].toInt();
	}
	auto v = ctx->get_saved_variables();
	auto g = RasterizeGaussiansBackwardCUDA(v[0] /*bg*/, v[5] /*means3D*/, v[9] /*radii*/, v[4] /*colors_precomp*/,
	                                        v[6] /*scales*/, v[7] /*rotations*/, scale_modifier, v[8] /*cov3Ds*/,
	                                        v[1] /*view*/, v[2] /*proj*/, tanfovx, tanfovy, grad_outputs[0], v[10] /*sh*/,
	                                        sh_degree, v[3] /*campos*/, v[11], num_rendered, v[12], v[13], raw_params,
	                                        sh_grad_view,
	                                        // view-factored mode: the SH step follows the exchange (gsr_sh_adam_from_views);
	                                        // sh_adam_ -- its lazy form -- served the forward pass (rows this view sees caught up)
	                                        // and lets backward run this step's slice of the rotating catch-up
	                                        bwd_adam, view_stats, geom_adam);
	// gradient order of the forward inputs (src/gaussian_rasterizer.cpp:159-179); absent optionals get none
	auto opt = [](const torch::Tensor& grad, const torch::Tensor& input) {
		return (input.defined() && input.numel() != 0 && grad.defined()) ? grad : torch::Tensor();
	};
	// (undefined where an extension took the gradient's place: fused optimizer steps, training_outputs_only)
	torch::autograd::tensor_list out = {std::get<3>(g) /*means3D*/,
	                                    std::get<0>(g) /*means2D*/,
	                                    opt(std::get<5>(g), v[10]) /*sh*/,
	                                    opt(std::get<1>(g), v[4]) /*colors_precomp*/,
	                                    std::get<2>(g) /*opacities*/,
	                                    opt(std::get<6>(g), v[6]) /*scales*/,
	                                    opt(std::get<7>(g), v[7]) /*rotations*/,
	                                    opt(std::get<4>(g), v[8]) /*cov3Ds_precomp*/};
	for (int i = 0; i < n_extra; i++) out.push_back(torch::Tensor());
	return out;
}

// GaussianRasterizer::forward, src/gaussian_rasterizer.cpp:182-234: the XOR validation and the empty optionals
void validate_and_fill(const torch::Tensor& means3D, bool has_shs, bool has_colors_precomp, bool has_scales, bool has_rotations,
                       bool has_cov3D_precomp, torch::Tensor& shs, torch::Tensor& colors_precomp, torch::Tensor& scales,
                       torch::Tensor& rotations, torch::Tensor& cov3D_precomp)
{
	if (has_shs == has_colors_precomp)
		throw std::runtime_error("Please provide excatly one of either SHs or precomputed colors!");
	if (((!has_scales || !has_rotations) && !has_cov3D_precomp) || ((has_scales || has_rotations) && has_cov3D_precomp))
		throw std::runtime_error(
		    "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
	auto empty = torch::empty({0}, means3D.options().dtype(torch::kFloat32));
	if (!has_shs) shs = empty;
	if (!has_colors_precomp) colors_precomp = empty;
	if (!has_scales) scales = empty;
	if (!has_rotations) rotations = empty;
	if (!has_cov3D_precomp) cov3D_precomp = empty;
}

}  // namespace

torch::autograd::tensor_list GaussianRasterizerFunction::forward(
    torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
    torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
    torch::Tensor cov3Ds_precomp, GaussianRasterizationSettings s)
{
	(void)means2D;  // only its gradient slot matters
	return forward_impl(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, s,
	                    GaussianRasterizationExtensions());
}

torch::autograd::tensor_list GaussianRasterizerFunction::backward(torch::autograd::AutogradContext* ctx,
                                                                  torch::autograd::tensor_list grad_outputs)
{
	return backward_impl(ctx, grad_outputs, 1);
}

torch::autograd::tensor_list GaussianRasterizerFunctionEx::forward(
    torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
    torch::Tensor colors_precomp, torch::Tensor opacities, torch::Tensor scales, torch::Tensor rotations,
    torch::Tensor cov3Ds_precomp, GaussianRasterizationSettings s, GaussianRasterizationExtensions e)
{
	(void)means2D;
	return forward_impl(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, s, e);
}

torch::autograd::tensor_list GaussianRasterizerFunctionEx::backward(torch::autograd::AutogradContext* ctx,
                                                                    torch::autograd::tensor_list grad_outputs)
{
	return backward_impl(ctx, grad_outputs, 2);
}

std::tuple<torch::Tensor, torch::Tensor> GaussianRasterizer::forward(
    torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, bool has_shs, bool has_colors_precomp,
    bool has_scales, bool has_rotations, bool has_cov3D_precomp, torch::Tensor shs, torch::Tensor colors_precomp,
    torch::Tensor scales, torch::Tensor rotations, torch::Tensor cov3D_precomp)
{
	validate_and_fill(means3D, has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp, shs, colors_precomp,
	                  scales, rotations, cov3D_precomp);
	auto result = rasterizeGaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
	                                 raster_settings_);
	return std::make_tuple(result[0], result[1]);
}

std::tuple<torch::Tensor, torch::Tensor> GaussianRasterizerEx::forward(
    torch::Tensor means3D, torch::Tensor means2D, torch::Tensor opacities, bool has_shs, bool has_colors_precomp,
    bool has_scales, bool has_rotations, bool has_cov3D_precomp, torch::Tensor shs, torch::Tensor colors_precomp,
    torch::Tensor scales, torch::Tensor rotations, torch::Tensor cov3D_precomp)
{
	validate_and_fill(means3D, has_shs, has_colors_precomp, has_scales, has_rotations, has_cov3D_precomp, shs, colors_precomp,
	                  scales, rotations, cov3D_precomp);
	auto result = GaussianRasterizerFunctionEx::apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
	                                                  cov3D_precomp, raster_settings_, extensions_);
	return std::make_tuple(result[0], result[1]);
}
