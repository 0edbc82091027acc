// gaussian_model_densify.cpp -- the map-maintenance half of GaussianModel (src/gaussian_model.cpp:114-191, 553-642,
// 716-815 of the reference): createFromPcd, oneUpShDegree, resetOpacity, prunePoints, densifyAndPrune.
//
// Same selection rules, same resulting order [originals that were not split | clones | split children] and the same
// Adam-state surgery as the reference, but every tensor is rebuilt ONCE per call: the reference copies each of the six
// parameter tensors and their moments 4-6 times (clone -> cat -> split -> cat -> prune -> prune, each through
// replaceTensorToOptimizer / catTensorstoOptimizer / prunePoints) and then empties the allocator cache.
#include <stdexcept>

#include "gaussian_model_lite.h"
#include "spatial.h"

namespace {
torch::Tensor inverse_sigmoid(const torch::Tensor& x) { return torch::log(x / (1 - x)); }   // include/general_utils.h
}  // namespace

GaussianModel::GaussianModel(int sh_degree) : max_sh_degree_(sh_degree), active_sh_degree_(0), spatial_lr_scale_(1.0f) {}

// src/gaussian_model.cpp:114-191: colours -> SH DC term (RGB2SH, include/sh_utils.h:138), scales from the mean squared
// distance to the three nearest neighbours (distCUDA2 = simple-knn), identity rotations, opacity 0.1.
void GaussianModel::createFromPcd(torch::Tensor points, torch::Tensor colors, float spatial_lr_scale)
{
	torch::NoGradGuard ng;
	spatial_lr_scale_ = spatial_lr_scale;
	auto pts = points.to(torch::kFloat32).contiguous();
	const auto n = pts.size(0);
	const auto o = pts.options();
	const double C0 = 0.28209479177387814;
	auto fused_color = (colors.to(o) - 0.5) / C0;
	const int64_t M = (max_sh_degree_ + 1) * (max_sh_degree_ + 1);
	auto features = torch::zeros({n, M, 3}, o);   // the reference's [n,3,M] transposed: one [n,M,3] leaf
	features.select(1, 0).copy_(fused_color);
	auto dist2 = torch::clamp_min(distCUDA2(pts), 0.0000001);
	auto scales = torch::log(torch::sqrt(dist2)).unsqueeze(1).repeat({1, 3});
	auto rots = torch::zeros({n, 4}, o);
	rots.select(1, 0).fill_(1.0);
	auto opacities = inverse_sigmoid(0.1 * torch::ones({n, 1}, o));
	auto leaf = [](torch::Tensor t) { return t.contiguous().set_requires_grad(true); };
	xyz_ = leaf(pts.clone());
	features_ = leaf(features);
	scaling_ = leaf(scales);
	rotation_ = leaf(rots);
	opacity_ = leaf(opacities);
	max_radii2D_ = torch::zeros({n}, o);
	xyz_gradient_accum_ = torch::zeros({n, 1}, o);
	denom_ = torch::zeros({n, 1}, o);
	groups_.clear();
}

void GaussianModel::oneUpShDegree()   // :72 / src/gaussian_model.cpp:98-102
{
	if (active_sh_degree_ < max_sh_degree_) active_sh_degree_++;
}

torch::Tensor& GaussianModel::paramByIndex(int i)
{
	switch (i) {
		case 0: return xyz_;
		case 1: return features_;
		case 2: return opacity_;
		case 3: return scaling_;
		default: return rotation_;
	}
}

// replaceTensorToOptimizer (src/gaussian_model.cpp:567-586) for the fused optimizer: the group keeps its
// hyper-parameters, the moments are the given tensors or zeros.
void GaussianModel::replaceParam(int group, torch::Tensor fresh, torch::Tensor exp_avg, torch::Tensor exp_avg_sq)
{
	fresh = fresh.contiguous().set_requires_grad(true);
	paramByIndex(group) = fresh;
	if (static_cast<size_t>(group) < groups_.size()) {
		auto& g = groups_[static_cast<size_t>(group)];
		g.param = fresh;
		g.exp_avg = exp_avg.defined() ? exp_avg : torch::zeros_like(fresh);
		g.exp_avg_sq = exp_avg_sq.defined() ? exp_avg_sq : torch::zeros_like(fresh);
	}
}

// src/gaussian_model.cpp:556-565 exactly as shipped: inverse_sigmoid(min(sigmoid(o), ones_like(sigmoid(o) * 0.01))) -- the
// 0.01 sits INSIDE ones_like, so the clamp is against 1 and never binds: the values survive (up to the sigmoid / logit round
// trip) and only the Adam moments of the opacity group are zeroed.  intended_opacity_reset_ (default off) selects the reset
// 3DGS intended, opacity <- min(opacity, 0.01): a deliberate deviation a caller has to ask for.
void GaussianModel::resetOpacity()
{
	torch::NoGradGuard ng;
	auto act = getOpacityActivation();
	auto bound = intended_opacity_reset_ ? torch::ones_like(act) * 0.01 : torch::ones_like(act * 0.01);
	auto fresh = inverse_sigmoid(torch::min(act, bound)).detach().clone();
	replaceParam(2, fresh, torch::Tensor(), torch::Tensor());
}

// One gather per tensor.  gather_index >= 0: existing row (keeps its Adam moments); < 0: copy of row (-1 - value) with
// zero moments; child_pos / child_xyz / child_scaling overwrite the rows of the split children.
void GaussianModel::rebuildWithSources(const torch::Tensor& gather_index, const torch::Tensor& child_pos,
                                       const torch::Tensor& child_xyz, const torch::Tensor& child_scaling)
{
	torch::NoGradGuard ng;
	auto new_rows = gather_index < 0;
	auto src = torch::where(new_rows, -1 - gather_index, gather_index);
	for (int i = 0; i < 5; i++) {
		auto old = paramByIndex(i).detach();
		auto fresh = old.index_select(0, src);
		if (child_pos.defined() && child_pos.numel()) {
			if (i == 0) fresh.index_copy_(0, child_pos, child_xyz);
			if (i == 3) fresh.index_copy_(0, child_pos, child_scaling);
		}
		torch::Tensor m, v;
		if (static_cast<size_t>(i) < groups_.size()) {
			m = groups_[static_cast<size_t>(i)].exp_avg.index_select(0, src);
			v = groups_[static_cast<size_t>(i)].exp_avg_sq.index_select(0, src);
			m.index_put_({new_rows}, 0.0f);
			v.index_put_({new_rows}, 0.0f);
		}
		replaceParam(i, fresh, m, v);
	}
}

// src/gaussian_model.cpp:588-642
void GaussianModel::prunePoints(torch::Tensor& mask)
{
	torch::NoGradGuard ng;
	auto keep = torch::nonzero(~mask).squeeze(1);
	rebuildWithSources(keep, torch::Tensor(), torch::Tensor(), torch::Tensor());
	xyz_gradient_accum_ = xyz_gradient_accum_.index_select(0, keep);
	denom_ = denom_.index_select(0, keep);
	max_radii2D_ = max_radii2D_.index_select(0, keep);
}

// src/gaussian_model.cpp:795-815 with densifyAndClone (:763-793), densifyAndSplit (:716-761, N = 2) and the final
// prunePoints (:805-813) folded into one rebuild.
GaussianModel::DensifyResult GaussianModel::densifyAndPrune(float max_grad, float min_opacity, float extent,
                                                            int max_screen_size, c10::optional<at::Generator> generator)
{
	torch::NoGradGuard ng;
	const int N = 2;
	auto grads = xyz_gradient_accum_ / denom_;
	grads.index_put_({grads.isnan()}, 0.0f);
	auto g = grads.squeeze(-1);
	auto scal = getScalingActivation().detach();
	auto smax = std::get<0>(scal.max(1));
	auto big = smax > opt_.percent_dense_ * extent;
	auto clone_mask = (g.abs() >= max_grad) & ~big;   // frobenius_norm over the last dimension of [P,1]
	auto split_mask = (g >= max_grad) & big;
	const auto P = xyz_.size(0);
	auto ar = torch::arange(P, torch::TensorOptions().dtype(torch::kLong).device(xyz_.device()));
	auto keep_idx = ar.index({~split_mask}), clone_idx = ar.index({clone_mask}), split_idx = ar.index({split_mask});
	auto rep = split_idx.repeat({N});
	// children: position sampled from the parent Gaussian, scale / (0.8 N)
	auto stds = scal.index_select(0, rep);
	auto samples = at::normal(torch::zeros_like(stds), stds, generator);
	auto q = rotation_.detach().index_select(0, rep);
	q = q / q.norm(2, {1}, true);
	auto r = q.select(1, 0), x = q.select(1, 1), y = q.select(1, 2), z = q.select(1, 3);
	auto R = torch::stack({1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
	                       2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
	                       2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}, 1).reshape({-1, 3, 3});
	auto child_xyz = torch::bmm(R, samples.unsqueeze(-1)).squeeze(-1) + xyz_.detach().index_select(0, rep);
	auto child_scaling = torch::log(stds / (0.8 * N));
	auto index = torch::cat({keep_idx, clone_idx, rep});
	auto is_new = torch::cat({torch::zeros_like(keep_idx, torch::kBool), torch::ones_like(clone_idx, torch::kBool),
	                          torch::ones_like(rep, torch::kBool)});
	// the final prune (:805-813), evaluated on the would-be tensors
	auto opac = torch::sigmoid(opacity_.detach().index_select(0, index)).squeeze(-1);
	auto prune = opac < min_opacity;
	if (max_screen_size) {
		// max_radii2D is reset by densificationPostfix before the prune, so big_points_vs is always false there
		auto new_smax = torch::cat({smax.index_select(0, keep_idx), smax.index_select(0, clone_idx),
		                            std::get<0>(torch::exp(child_scaling).max(1))});
		prune = prune | (new_smax > 0.1f * extent);   // float product, as the reference
	}
	auto sel = ~prune;
	const auto n_old = keep_idx.size(0) + clone_idx.size(0);
	auto child_pos_all = torch::arange(n_old, index.size(0), ar.options());
	auto new_pos = torch::cumsum(sel.to(torch::kLong), 0) - 1;
	auto child_sel = sel.index_select(0, child_pos_all);
	auto child_pos = new_pos.index_select(0, child_pos_all).index({child_sel});
	auto index_f = index.index({sel});
	auto is_new_f = is_new.index({sel});
	auto gather_index = torch::where(is_new_f, -1 - index_f, index_f);
	rebuildWithSources(gather_index, child_pos, child_xyz.index({child_sel}), child_scaling.index({child_sel}));
	const auto n = gather_index.size(0);
	xyz_gradient_accum_ = torch::zeros({n, 1}, xyz_.options().requires_grad(false));
	denom_ = torch::zeros({n, 1}, xyz_.options().requires_grad(false));
	max_radii2D_ = torch::zeros({n}, xyz_.options().requires_grad(false));
	DensifyResult res;
	res.cloned = clone_idx.size(0);
	res.split = split_idx.size(0);
	res.pruned = prune.sum().item<int64_t>();
	res.points = n;
	return res;
}
