// train_step.cpp -- LibTorch host code of the measured train step (see gaussian_model_lite.h).
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../../include/gsr.h"
#include "gaussian_model_lite.h"
#include "gaussian_renderer.h"
#include "loss_utils.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif

namespace {
void* stream_of(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}
void check(int status, const char* where)
{
	if (status != GSR_OK) throw std::runtime_error(std::string(where) + ": " + gsr_strerror(status));
}

}  // namespace

torch::Tensor fusedL1SSIMLoss(torch::Tensor rendered, torch::Tensor gt, torch::Tensor mask, float lambda_dssim, bool is_root)
{
	return loss_utils::fused_l1_ssim(rendered, gt, mask, lambda_dssim, is_root);   // (lib cuda_rasterizer: host/src/loss_utils.cpp)
}

GaussianModel::GaussianModel(int sh_degree, torch::Tensor xyz, torch::Tensor features, torch::Tensor opacity,
                             torch::Tensor scaling, torch::Tensor rotation, float spatial_lr_scale)
    : max_sh_degree_(sh_degree), active_sh_degree_(sh_degree), spatial_lr_scale_(spatial_lr_scale)
{
	auto leaf = [](torch::Tensor t) { return t.detach().clone().contiguous().set_requires_grad(true); };
	xyz_ = leaf(xyz);
	features_ = leaf(features);
	opacity_ = leaf(opacity);
	scaling_ = leaf(scaling);
	rotation_ = leaf(rotation);
	const auto P = xyz_.size(0);
	max_radii2D_ = torch::zeros({P}, xyz_.options());
	xyz_gradient_accum_ = torch::zeros({P, 1}, xyz_.options());
	denom_ = torch::zeros({P, 1}, xyz_.options());
	exist_since_iter_ = torch::zeros({P}, xyz_.options().dtype(torch::kInt32).requires_grad(false));
}

// src/gaussian_model.cpp:73-96: Sigma = (R S)(R S)^T with R = build_rotation(rotation_) (include/general_utils.h:33-57: the
// quaternion is normalised there), S = diag(scaling_modifier * exp(scaling_)); the six entries xx xy xz yy yz zz
torch::Tensor GaussianModel::getCovarianceActivation(int scaling_modifier)
{
	auto q = rotation_ / torch::sqrt((rotation_ * rotation_).sum(1, /*keepdim=*/true));
	auto r = q.select(1, 0), x = q.select(1, 1), y = q.select(1, 2), z = q.select(1, 3);
	auto R = torch::stack({1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
	                       2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
	                       2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}, 1).reshape({-1, 3, 3});
	auto L = R * (scaling_modifier * getScalingActivation()).unsqueeze(1);   // R diag(s): column j scaled by s_j
	auto cov = torch::bmm(L, L.transpose(1, 2));
	return torch::stack({cov.select(1, 0).select(1, 0), cov.select(1, 0).select(1, 1), cov.select(1, 0).select(1, 2),
	                     cov.select(1, 1).select(1, 1), cov.select(1, 1).select(1, 2), cov.select(1, 2).select(1, 2)}, 1);
}

void GaussianModel::trainingSetup(const GaussianOptimizationParams& opt)
{
	opt_ = opt;
	groups_.clear();
	auto add = [&](torch::Tensor& p, double lr, int period = 0, int split = 0, double lr_tail = 0.0) {
		AdamGroup g;
		g.param = p;
		g.exp_avg = torch::zeros_like(p);
		g.exp_avg_sq = torch::zeros_like(p);
		g.lr = lr;
		g.lr_tail = lr_tail;
		g.period = period;
		g.split = split;
		groups_.push_back(g);
	};
	add(xyz_, opt.position_lr_init_ * spatial_lr_scale_);
	// features_dc (lr) | features_rest (lr / 20) share the [P,16,3] buffer
	add(features_, opt.feature_lr_, 3 * (max_sh_degree_ + 1) * (max_sh_degree_ + 1), 3, opt.feature_lr_ / 20.0);   // double division, :495
	add(opacity_, opt.opacity_lr_);
	add(scaling_, opt.scaling_lr_);
	add(rotation_, opt.rotation_lr_);
	preloadMaintenanceKernels();
}

// The HIP runtime loads a code object of LibTorch the first time one of its kernels is launched.  The train step itself uses
// none of ATen's elementwise kernels, so the first resetOpacity (sigmoid, minimum) or loop-closure call (integer abs / compare)
// in the middle of a mapping session would pay for that: 25 ms on a warm box, 110 ms on a cold one, measured as ONE iteration of
// bench.py --mapper-loop (the second reset of the same run costs nothing measurable).  They are touched here, on four elements,
// when the optimizer is set up -- a session pays at its start, not at iteration opacity_reset_interval.
void GaussianModel::preloadMaintenanceKernels()
{
	if (!xyz_.defined() || !xyz_.is_cuda()) return;
	torch::NoGradGuard ng;
	const auto o = xyz_.options().requires_grad(false);
	auto x = torch::full({4, 1}, 0.25f, o);
	auto act = torch::sigmoid(x);
	auto bound = torch::ones_like(act * 0.01);
	auto y = torch::log(torch::min(act, bound) / (1 - torch::min(act, torch::ones_like(act) * 0.01)));   // resetOpacity, both forms
	auto age = torch::zeros({4}, o.dtype(torch::kInt32));
	auto young = torch::abs(age - 3) < 2;                                                               // scaledTransformVisiblePointsOfKeyframe
	auto z = (x * 1.5f).contiguous().clone().zero_();
	(void)y; (void)young; (void)z;
}

float GaussianModel::updateLearningRate(int step)
{
	const float lr_init = opt_.position_lr_init_ * spatial_lr_scale_, lr_final = opt_.position_lr_final_ * spatial_lr_scale_;
	float lr = 0.f;
	if (!(step < 0 || (lr_init == 0.0f && lr_final == 0.0f))) {
		float t = static_cast<float>(step) / static_cast<float>(opt_.position_lr_max_steps_);
		t = std::min(std::max(t, 0.0f), 1.0f);
		lr = std::exp(std::log(lr_init) * (1 - t) + std::log(lr_final) * t);
	}
	groups_[0].lr = lr;
	return lr;
}

void GaussianModel::syncFeatures()
{
	if (!features_row_step_.defined()) return;
	torch::NoGradGuard ng;
	auto row_step = features_row_step_;
	features_row_step_ = torch::Tensor();   // (first: the calls below may come back here)
	if (!features_lr_hist_.empty() && groups_.size() > 1) {
		auto& grp = groups_[1];
		ShAdamStep s;
		s.exp_avg = grp.exp_avg;
		s.exp_avg_sq = grp.exp_avg_sq;
		s.step = grp.step;   // the steps the tensor has taken; the newest entry of the history belongs to it
		s.lr = features_lr_hist_[0].first;
		s.lr_tail = features_lr_hist_[0].second;
		s.row_step = row_step;
		s.window = features_lazy_window_;
		for (size_t k = 1; k < features_lr_hist_.size(); k++) {
			s.lr_past.push_back(features_lr_hist_[k].first);
			s.lr_tail_past.push_back(features_lr_hist_[k].second);
		}
		auto sh = features_.detach();
		shAdamFlush(sh, s);
	}
	features_lr_hist_.clear();
}

void GaussianModel::optimizerStepGroup(int group)
{
	torch::NoGradGuard ng;
	auto& g = groups_.at(static_cast<size_t>(group));
	auto grad = g.param.grad();
	if (!grad.defined()) return;
	if (group == 1) syncFeatures();   // a dense step of the SH tensor: every row must be up to date first
	grad = grad.contiguous();
	g.step++;
	check(gsr_adam_step(g.param.data_ptr<float>(), grad.data_ptr<float>(), g.exp_avg.data_ptr<float>(),
	                    g.exp_avg_sq.data_ptr<float>(), g.param.numel(), g.lr * lr_scale_, 0.9, 0.999, 1e-15, g.step, g.period,
	                    g.split, (g.period ? g.lr_tail : g.lr) * lr_scale_, stream_of(g.param)),
	      "gsr_adam_step");
}

void GaussianModel::optimizerStep()
{
	beginOptimizerStep();
	for (int i = 0; i < static_cast<int>(groups_.size()); i++) optimizerStepGroup(i);
}

void GaussianModel::zeroGrad()
{
	for (auto& g : groups_) g.param.mutable_grad() = torch::Tensor();
}

void GaussianModel::addDensificationStats(torch::Tensor& viewspace_point_tensor, torch::Tensor& update_filter)
{
	auto g = viewspace_point_tensor.grad();
	auto n = torch::norm(g.index({update_filter}).slice(1, 0, 2), 2, {-1}, true);
	xyz_gradient_accum_.index_put_({update_filter}, xyz_gradient_accum_.index({update_filter}) + n);
	denom_.index_put_({update_filter}, denom_.index({update_filter}) + 1);
}

torch::Tensor TrainStep::renderAndBackward(std::shared_ptr<GaussianKeyframe> kf, torch::Tensor gt_image, torch::Tensor mask)
{
	auto& g = gaussians_;
	iteration_++;
	// the position learning rate follows the iteration (src/gaussian_mapper.cpp:672-674) or -- a SLAM session -- the number of
	// times THIS keyframe has been used (:663-671): the caller says which through position_lr_step_
	g->updateLearningRate(position_lr_step_ >= 0 ? std::min(position_lr_step_, g->opt_.position_lr_max_steps_) : iteration_);
	GaussianPipelineParams& pipe = pipe_;
	torch::Tensor override_color;
	if (factored_exchange_) {
		const auto P = g->xyz_.size(0);
		sh_send_ = torch::empty({P + 1, 3}, g->xyz_.options().requires_grad(false));
		sh_grad_view_ = sh_send_.narrow(0, 0, P);
		sh_send_.select(0, P).copy_(kf->camera_center_.detach().reshape({3}));
		// the packed form needs the fused [P,16,3] step of the views (the other layouts take the dense route)
		packed_this_step_ = packed_exchange_ && process_group_ && g->features_.size(1) == 16 && g->groups_.size() > 1 &&
		                    g->features_.is_contiguous() && !pipe_.convert_SHs_;
		if (process_group_ && !packed_this_step_) {
			const int64_t N = process_group_->getSize();
			if (!sh_gathered_.defined() || sh_gathered_.size(0) != N || sh_gathered_.size(1) != P + 1 ||
			    sh_gathered_.device() != sh_send_.device())
				sh_gathered_ = torch::empty({N, P + 1, 3}, sh_send_.options());
		}
		if (packed_this_step_) {
			// persistent message buffers sized for the worst case (every Gaussian visible), allocated HERE -- before the passes
			// are enqueued, like sh_gathered_
			const int64_t N = process_group_->getSize(), words = packedViewWords(P, (P + 3) / 4 * 4);
			const auto io = sh_send_.options().dtype(torch::kInt32);
			if (!sh_packed_send_.defined() || sh_packed_send_.numel() != words || sh_packed_send_.device() != sh_send_.device()) {
				sh_packed_send_ = torch::empty({words}, io);
				sh_packed_gathered_ = torch::empty({N * words}, io);
				sh_gathered_ = torch::Tensor();
			}
			if (sh_packed_gathered_.numel() != N * words) sh_packed_gathered_ = torch::empty({N * words}, io);
		}
	} else {
		sh_send_ = torch::Tensor();
		sh_grad_view_ = torch::Tensor();
		sh_gathered_ = torch::Tensor();
		packed_this_step_ = false;
	}
	ShAdamStep sh_adam;
	const auto& o = g->opt_;
	const bool rebuilds = densifyDue();
	bool lazy = false;
	// The fused optimizer steps advance their step counters HERE, so they are taken only when render() will really hand them
	// to the rasterizer (gaussian_renderer.h: sh_in_rasterizer, raw_params_ == 7): with convert_SHs_ the rasterizer sees colours,
	// not the SH tensor, and with compute_cov3D_ a covariance, not the raw scaling / rotation leaves -- autograd then leaves
	// dense gradients and optimizerStepGroup() takes those groups' steps.
	if (fused_sh_adam_ && !factored_exchange_ && !rebuilds && iteration_ < o.iterations_ && g->groups_.size() > 1 &&
	    g->features_.size(1) == 16 && !pipe.convert_SHs_) {
		auto& grp = g->groups_[1];
		lazy = lazy_sh_adam_window_ >= 2 && g->features_.is_contiguous();
		if (lazy && g->features_row_step_.defined() && g->features_lazy_window_ != lazy_sh_adam_window_) g->syncFeatures();
		if (lazy && !g->features_row_step_.defined()) {   // every row has taken the grp.step steps so far
			g->features_row_step_ = torch::full({g->features_.size(0)}, grp.step, g->features_.options().dtype(torch::kInt32).requires_grad(false));
			g->features_lazy_window_ = lazy_sh_adam_window_;
			g->features_lr_hist_.clear();
		}
		grp.step++;   // the step happens inside backward; optimizerStepGroup(1) then finds no gradient
		sh_adam.exp_avg = grp.exp_avg;
		sh_adam.exp_avg_sq = grp.exp_avg_sq;
		sh_adam.lr = grp.lr * g->lr_scale_;
		sh_adam.lr_tail = grp.lr_tail * g->lr_scale_;
		sh_adam.step = grp.step;
		if (lazy) {
			sh_adam.row_step = g->features_row_step_;
			sh_adam.window = g->features_lazy_window_;
			for (const auto& h : g->features_lr_hist_) {
				sh_adam.lr_past.push_back(h.first);
				sh_adam.lr_tail_past.push_back(h.second);
			}
		}
	}
	// Data-parallel step with the view-factored exchange: the SH rows step AFTER the exchange (stepFeaturesFromViews), and lazily
	// there too -- a row no view of the batch lights takes a zero-gradient step, i.e. it may take it later.  The forward pass
	// gets the same struct, so that the rows THIS view sees are up to date before they are evaluated; backward (factored mode)
	// uses it only to run this step's slice of the rows' rotating catch-up next to the blend kernel.
	views_adam_ = ShAdamStep();
	views_adam_pending_ = false;
	if (factored_exchange_ && lazy_sh_adam_window_ >= 3 && !rebuilds && iteration_ < o.iterations_ && g->groups_.size() > 1 &&
	    g->features_.size(1) == 16 && g->features_.is_contiguous() && !pipe.convert_SHs_) {
		auto& grp = g->groups_[1];
		lazy = true;
		if (g->features_row_step_.defined() && g->features_lazy_window_ != lazy_sh_adam_window_) g->syncFeatures();
		if (!g->features_row_step_.defined()) {   // every row has taken the grp.step steps so far
			g->features_row_step_ = torch::full({g->features_.size(0)}, grp.step, g->features_.options().dtype(torch::kInt32).requires_grad(false));
			g->features_lazy_window_ = lazy_sh_adam_window_;
			g->features_lr_hist_.clear();
		}
		grp.step++;   // the step happens in stepFeaturesFromViews(); optimizerStepGroup(1) finds no gradient
		sh_adam.exp_avg = grp.exp_avg;
		sh_adam.exp_avg_sq = grp.exp_avg_sq;
		sh_adam.lr = grp.lr * g->lr_scale_;
		sh_adam.lr_tail = grp.lr_tail * g->lr_scale_;
		sh_adam.step = grp.step;
		sh_adam.row_step = g->features_row_step_;
		sh_adam.window = g->features_lazy_window_;
		for (const auto& h : g->features_lr_hist_) {
			sh_adam.lr_past.push_back(h.first);
			sh_adam.lr_tail_past.push_back(h.second);
		}
		views_adam_ = sh_adam;
		views_adam_pending_ = true;
	}
	if (!lazy) g->syncFeatures();   // the render below reads every visible row as it is
#ifndef GSR_HOST_NO_HIP
	// early_gather_ (GSR_EARLY_GATHER=0/1 overrides): off = the gather is issued on the compute stream behind the whole backward pass
	static const int early_env = [] { const char* e = getenv("GSR_EARLY_GATHER"); return (e && *e) ? (atoi(e) != 0 ? 1 : 0) : -1; }();
	const bool early_gather = early_env >= 0 ? early_env != 0 : early_gather_;
	gather_stream_in_use_ = false;
	if (early_gather && factored_exchange_ && process_group_ && g->xyz_.is_cuda()) {
		// the exchange's gather waits for the colour gradients only, not for the whole backward pass (keyframe_batch_exchange.cpp)
		if (!gather_stream_) gather_stream_ = c10::hip::getStreamFromPool(/*isHighPriority=*/false, g->xyz_.device().index()).stream();
		sh_adam.color_view_ready_stream = gather_stream_;
		gather_stream_in_use_ = true;
	}
#endif
	sh_adam.no_side_stream = no_side_stream_;
	sh_adam.lazy_slice_late = lazy_slice_late_;
	// packed exchange: the backward pass writes this view's message itself (rows + header; the mask / prefix sections are
	// planned from the radii right behind the forward pass, below) -- no pack launches between the backward pass and the gather
	static const int pack_env = [] { const char* e = getenv("GSR_PACK_IN_BACKWARD"); return (e && *e) ? (atoi(e) != 0 ? 1 : 0) : -1; }();
	prepacked_this_step_ = packed_this_step_ && (pack_env >= 0 ? pack_env != 0 : pack_in_backward_);
	if (prepacked_this_step_) {
		sh_adam.packed_view = sh_packed_send_;
		sh_adam.packed_capacity = (g->xyz_.size(0) + 3) / 4 * 4;
	}
	GeomAdamStep geom_adam;
	// (an iteration that resets the opacity replaces that leaf AFTER backward: the reference's optimizer step then skips it -- no
	// gradient -- while a step fused into backward would already have been taken: src/gaussian_mapper.cpp:732-735)
	const bool resets = densify_ && iteration_ < o.densify_until_iter_ && o.opacity_reset_interval_ &&
	                    iteration_ % o.opacity_reset_interval_ == 0;
	if (fused_geom_adam_ && !factored_exchange_ && sh_adam.exp_avg.defined() && g->groups_.size() == 5 && !pipe.compute_cov3D_ && !resets) {
		// xyz, opacity, scaling, rotation = groups 0, 2, 3, 4 (trainingSetup); their steps happen inside backward, and
		// optimizerStepGroup() then finds no gradient on them
		for (int gi : {0, 2, 3, 4}) {
			auto& grp = g->groups_[static_cast<size_t>(gi)];
			grp.step++;
			geom_adam.param.push_back(grp.param.detach());
			geom_adam.exp_avg.push_back(grp.exp_avg);
			geom_adam.exp_avg_sq.push_back(grp.exp_avg_sq);
			geom_adam.lr.push_back(grp.lr * g->lr_scale_);
			geom_adam.step.push_back(grp.step);
		}
	}
	// the statistics are fused (or over) in every mode of this step: nobody reads the viewspace gradient or dL_dcov3D
	geom_adam.training_outputs_only = true;
	// the densification statistics of this view (:714-719) are added by the backward kernel that holds dL_dmean2D in
	// registers
	std::vector<torch::Tensor> view_stats;
	if (iteration_ < o.densify_until_iter_) view_stats = {g->xyz_gradient_accum_, g->denom_, g->max_radii2D_};
	g->in_lazy_step_ = lazy;
	auto pkg = GaussianRenderer::render(kf, kf->image_height_, kf->image_width_, g, pipe, background_, override_color,
	                                    1.0f, false, /*fuse_activations=*/true, sh_grad_view_, sh_adam, view_stats, geom_adam,
	                                    cull_empty_tiles_, persistent_workspace_ ? &workspace_ : nullptr);
	g->in_lazy_step_ = false;
	if (factored_exchange_ && packed_this_step_) beginCountExchange();   // (the forward pass has left this view's visible count)
	if (prepacked_this_step_) planPackedView(std::get<3>(pkg));
	auto rendered = std::get<0>(pkg);
	last_viewspace_ = std::get<1>(pkg);
	last_visibility_ = std::get<2>(pkg);
	last_radii_ = std::get<3>(pkg);
	// rendered * mask with a mask of ones is the identity (src/gaussian_mapper.cpp:692-693; most keyframes carry a full mask):
	// recognised ONCE per mask tensor (one reduction + host read when a keyframe's mask is first seen, remembered by storage
	// pointer and version counter); the loss kernels then skip its 2 x 25 MB of reads at 1080p
	torch::Tensor eff_mask = mask;
	if (mask.defined() && mask.numel()) {
		// remembered per TensorImpl through a weak pointer (it keeps the object's address from being reused by another tensor
		// after this one is freed) together with the version counter (in-place writes invalidate the entry)
		auto* impl = mask.unsafeGetTensorImpl();
		const int64_t version = static_cast<int64_t>(mask._version());
		auto it = mask_is_ones_.find(impl);
		if (it == mask_is_ones_.end() || it->second.self.expired() || it->second.version != version) {
			if (mask_is_ones_.size() >= 64) mask_is_ones_.clear();
			torch::NoGradGuard ng;
			MaskEntry e;
			e.self = c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>(mask.getIntrusivePtr());
			e.version = version;
			e.ones = (mask == 1).all().item<bool>();
			it = mask_is_ones_.insert_or_assign(impl, std::move(e)).first;
		}
		if (it->second.ones) eff_mask = torch::empty({0}, mask.options());   // (an empty mask = none: FusedL1SSIMFunction::forward)
	}
	auto loss = fusedL1SSIMLoss(rendered, gt_image, eff_mask, g->opt_.lambda_dssim_, /*is_root=*/true);
	// the root gradient: a cached 1 instead of the ones_like fill autograd launches per backward()
	if (!root_grad_.defined() || root_grad_.device() != loss.device()) root_grad_ = torch::ones_like(loss).detach();
	loss.backward(root_grad_);
	if (lazy && !views_adam_pending_) {   // the step is taken: its learning rates join the history the later catch-ups need
		auto& hist = g->features_lr_hist_;
		hist.insert(hist.begin(), {sh_adam.lr, sh_adam.lr_tail});
		if (static_cast<int>(hist.size()) > g->features_lazy_window_) hist.pop_back();
	}
	return loss;
}

void TrainStep::finishOneIteration()
{
	finishBegin();
	if (iteration_ < gaussians_->opt_.iterations_)
		for (int i = 0; i < static_cast<int>(gaussians_->groups_.size()); i++) gaussians_->optimizerStepGroup(i);
	finishEnd();
}

void TrainStep::setFeaturesGradFromViews(torch::Tensor campos_views, torch::Tensor dL_dcolor_views)
{
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	if (views_adam_pending_) {
		// the dense route after all: the lazy step renderAndBackward() announced is withdrawn (its counter, not yet its learning
		// rates, had been recorded) and every row catches up to the steps really taken
		views_adam_pending_ = false;
		views_adam_ = ShAdamStep();
		g->groups_[1].step--;
	}
	g->syncFeatures();
	g->features_.mutable_grad() =
	    shGradFromViews(g->xyz_.detach(), campos_views, dL_dcolor_views, g->active_sh_degree_,
	                    static_cast<int>(g->features_.size(1)), 1.0f / static_cast<float>(dL_dcolor_views.size(0)));
}

void TrainStep::stepFeaturesFromViews(torch::Tensor campos_views, torch::Tensor dL_dcolor_views, int64_t row0, bool first_part)
{
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	if (iteration_ >= g->opt_.iterations_) return;
	const int64_t P = g->xyz_.size(0), n = dL_dcolor_views.size(1);
	if (row0 < 0 || row0 + n > P) throw std::runtime_error("stepFeaturesFromViews: the part exceeds the Gaussians");
	if (views_adam_pending_) {
		// lazy rows: renderAndBackward() advanced the step counter and prepared the struct; rows no view lights stay behind
		ShAdamStep a = views_adam_;
		a.exp_avg = views_adam_.exp_avg.narrow(0, row0, n);
		a.exp_avg_sq = views_adam_.exp_avg_sq.narrow(0, row0, n);
		a.row_step = views_adam_.row_step.narrow(0, row0, n);
		auto sh = g->features_.detach().narrow(0, row0, n);
		shAdamFromViews(g->xyz_.detach().narrow(0, row0, n), campos_views, dL_dcolor_views, g->active_sh_degree_,
		                1.0f / static_cast<float>(dL_dcolor_views.size(0)), sh, a);
		return;
	}
	g->syncFeatures();
	if (g->features_.size(1) != 16 || g->groups_.size() < 2) {   // other layouts: gradient tensor + separate pass (whole batch only)
		if (row0 != 0 || n != P) throw std::runtime_error("stepFeaturesFromViews: parts need the [P,16,3] SH layout");
		setFeaturesGradFromViews(campos_views, dL_dcolor_views);
		finishAdamGroup(1);
		return;
	}
	auto& grp = g->groups_[1];
	if (first_part) grp.step++;
	ShAdamStep a;
	a.exp_avg = grp.exp_avg.narrow(0, row0, n);
	a.exp_avg_sq = grp.exp_avg_sq.narrow(0, row0, n);
	a.lr = grp.lr * g->lr_scale_;
	a.lr_tail = grp.lr_tail * g->lr_scale_;
	a.step = grp.step;
	auto sh = g->features_.detach().narrow(0, row0, n);
	shAdamFromViews(g->xyz_.detach().narrow(0, row0, n), campos_views, dL_dcolor_views, g->active_sh_degree_,
	                1.0f / static_cast<float>(dL_dcolor_views.size(0)), sh, a);
}

void TrainStep::stepFeaturesFromPackedViews(torch::Tensor messages, int64_t msg_stride, int64_t n_views)
{
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	if (iteration_ >= g->opt_.iterations_) return;
	const float scale = 1.0f / static_cast<float>(n_views);
	auto sh = g->features_.detach();
	if (views_adam_pending_) {   // lazy rows: renderAndBackward() advanced the step counter and prepared the struct
		shAdamFromPackedViews(g->xyz_.detach(), messages, msg_stride, n_views, g->active_sh_degree_, scale, sh, views_adam_);
		return;
	}
	g->syncFeatures();
	auto& grp = g->groups_[1];
	grp.step++;
	ShAdamStep a;
	a.exp_avg = grp.exp_avg;
	a.exp_avg_sq = grp.exp_avg_sq;
	a.lr = grp.lr * g->lr_scale_;
	a.lr_tail = grp.lr_tail * g->lr_scale_;
	a.step = grp.step;
	shAdamFromPackedViews(g->xyz_.detach(), messages, msg_stride, n_views, g->active_sh_degree_, scale, sh, a);
}

void TrainStep::finishFeaturesFromViews()
{
	if (!views_adam_pending_) return;
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	views_adam_pending_ = false;
	// (the rotating catch-up that bounds the lag of the rows no view lights ran inside the rasterizer's backward, next to the
	// blend kernel: gsr_backward_args.sh_adam together with dL_dcolor_view)
	auto& hist = g->features_lr_hist_;   // the step is taken: its learning rates join the history the later catch-ups need
	hist.insert(hist.begin(), {views_adam_.lr, views_adam_.lr_tail});
	if (static_cast<int>(hist.size()) > g->features_lazy_window_) hist.pop_back();
	views_adam_ = ShAdamStep();
}

void TrainStep::finishGeomAdam(float grad_scale)
{
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	if (iteration_ >= g->opt_.iterations_) return;
	if (g->groups_.size() != 5) {
		// another parameter layout than trainingSetup()'s five groups: every group but the SH tensor's (stepped from the views)
		// takes its own pass, the 1/N of the summed gradients applied first -- never a silent return: in the factored
		// data-parallel step this function is the ONLY consumer of those gradients
		for (int gi = 0; gi < static_cast<int>(g->groups_.size()); gi++) {
			if (gi == 1) continue;
			auto grad = g->groups_[static_cast<size_t>(gi)].param.grad();
			if (!grad.defined()) continue;
			if (grad_scale != 1.0f) grad.mul_(grad_scale);
			finishAdamGroup(gi);
		}
		return;
	}
	std::vector<AdamMultiEntry> entries;
	for (int gi : {0, 2, 3, 4}) {
		auto& grp = g->groups_[static_cast<size_t>(gi)];
		auto grad = grp.param.grad();
		if (!grad.defined()) continue;   // (the iteration that rebuilt the tensors: no gradient, the step counter rests)
		grp.step++;
		AdamMultiEntry e;
		e.param = grp.param.detach();
		e.grad = grad.contiguous();
		e.exp_avg = grp.exp_avg;
		e.exp_avg_sq = grp.exp_avg_sq;
		e.lr = grp.lr * g->lr_scale_;
		e.step = grp.step;
		e.grad_scale = grad_scale;
		entries.push_back(e);
	}
	adamStepMulti(entries, 0.9, 0.999, 1e-15);
	for (int gi : {0, 2, 3, 4}) g->groups_[static_cast<size_t>(gi)].param.mutable_grad() = torch::Tensor();
}

void TrainStep::finishAdamGroup(int group)
{
	if (iteration_ < gaussians_->opt_.iterations_) gaussians_->optimizerStepGroup(group);
}

void TrainStep::finishEnd()
{
	finishFeaturesFromViews();   // (a driver that forgot the slice: the lazy state stays consistent)
	if (iteration_ < gaussians_->opt_.iterations_) gaussians_->zeroGrad();
}

bool TrainStep::densifyDue() const
{
	const auto& o = gaussians_->opt_;
	return densify_ && iteration_ < o.densify_until_iter_ && iteration_ > o.densify_from_iter_ && o.densification_interval_ &&
	       iteration_ % o.densification_interval_ == 0;
}

void TrainStep::finishBegin()
{
	torch::NoGradGuard ng;
	auto& g = gaussians_;
	// (the statistics of :714-719 were added inside backward: view_stats)
	if (iteration_ < g->opt_.densify_until_iter_ && densify_) {
		const auto& o = g->opt_;
		if (densifyDue()) {
			const int size_threshold = iteration_ > prune_big_point_after_iter_ ? 20 : 0;   // src/gaussian_mapper.cpp:723
			g->zeroGrad();   // shapes change; this step's update is skipped
			last_densify_ = g->densifyAndPrune(o.densify_grad_threshold_, densify_min_opacity_, cameras_extent_, size_threshold,
			                                   generator_);
		}
		if (o.opacity_reset_interval_ && iteration_ % o.opacity_reset_interval_ == 0) g->resetOpacity();
	}
	if (iteration_ < g->opt_.iterations_) g->beginOptimizerStep();
}
