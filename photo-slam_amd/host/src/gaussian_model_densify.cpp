// gaussian_model_densify.cpp -- the map-maintenance half of GaussianModel (src/gaussian_model.cpp:114-191, 553-642,
// 716-815 of the reference): createFromPcd, oneUpShDegree, resetOpacity, prunePoints, densifyAndPrune.
//
// Same selection rules, same resulting order [originals that were not split | clones | split children] and the same
// Adam-state surgery as the reference, but every tensor is rebuilt ONCE per call: the reference copies each of the six
// parameter tensors and their moments 4-6 times (clone -> cat -> split -> cat -> prune -> prune, each through
// replaceTensorToOptimizer / catTensorstoOptimizer / prunePoints) and then empties the allocator cache.
#include <array>
#include <stdexcept>
#include <string>

#include "../../../include/gsr.h"
#include "gaussian_model_lite.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif
#include "spatial.h"
#include "operate_points.h"

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
	// RGB2SH, include/sh_utils.h:138: (rgb - 0.5f) / C0 with the FLOAT constant as a host scalar, exactly the reference's
	// expression: ATen divides on the host and multiplies by the reciprocal of THAT float on the GPU (the double
	// 0.28209479177387814 has another reciprocal in float: one ulp off in nearly every row, found on the GPU box)
	const float C0 = 0.28209479177387814f;
	auto fused_color = (colors.to(o) - 0.5f) / C0;
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
	features_row_step_ = torch::Tensor();   // a new SH tensor: no lazy state
	features_lr_hist_.clear();
	features_ = leaf(features);
	scaling_ = leaf(scales);
	rotation_ = leaf(rots);
	opacity_ = leaf(opacities);
	max_radii2D_ = torch::zeros({n}, o);
	xyz_gradient_accum_ = torch::zeros({n, 1}, o);
	denom_ = torch::zeros({n, 1}, o);
	exist_since_iter_ = torch::zeros({n}, o.dtype(torch::kInt32));   // :167-169
	groups_.clear();
}

// src/gaussian_model.cpp:193-290: the std::vector overload wraps the floats and proceeds as the tensor overload (:292-376)
void GaussianModel::increasePcd(std::vector<float> points, std::vector<float> colors, const int iteration)
{
	if (points.size() != colors.size() || points.size() % 3 != 0) throw std::runtime_error("increasePcd: points / colors must hold 3 floats per point");
	const int64_t n = static_cast<int64_t>(points.size() / 3);
	if (n == 0) return;
	const auto dev = xyz_.defined() ? xyz_.device() : device_;
	auto p = torch::from_blob(points.data(), {n, 3}, torch::kFloat32).clone().to(dev);
	auto c = torch::from_blob(colors.data(), {n, 3}, torch::kFloat32).clone().to(dev);
	increasePcd(p, c, iteration);
}

void GaussianModel::increasePcd(torch::Tensor& new_point_cloud, torch::Tensor& new_colors, const int iteration)
{
	torch::NoGradGuard ng;
	const int64_t n = new_point_cloud.size(0);
	if (n == 0) return;
	const auto o = xyz_.options().requires_grad(false);
	auto pts = new_point_cloud.to(o).contiguous(), cols = new_colors.to(o).contiguous();
	if (!sparse_points_xyz_.defined() || sparse_points_xyz_.size(0) == 0) {   // :205-212
		sparse_points_xyz_ = pts;
		sparse_points_color_ = cols;
	} else {
		sparse_points_xyz_ = torch::cat({sparse_points_xyz_, pts}, 0);
		sparse_points_color_ = torch::cat({sparse_points_color_, cols}, 0);
	}
	const float C0 = 0.28209479177387814f;   // (the FLOAT constant: see createFromPcd)
	const int64_t M = (max_sh_degree_ + 1) * (max_sh_degree_ + 1);
	auto features = torch::zeros({n, M, 3}, o);
	features.select(1, 0).copy_((cols - 0.5f) / C0);   // RGB2SH, include/sh_utils.h:138
	auto dist2 = torch::clamp_min(distCUDA2(pts.clone()), 0.0000001);
	auto scales = torch::log(torch::sqrt(dist2)).unsqueeze(1).repeat({1, 3});
	auto rots = torch::zeros({n, 4}, o);
	rots.select(1, 0).fill_(1.0);
	auto opacities = inverse_sigmoid(0.1 * torch::ones({n, 1}, o));
	appendRows({pts, features, opacities, scales, rots}, iteration);
}

// densificationPostfix (src/gaussian_model.cpp:644-712) as an append: while the arena has room and the live tensors are its
// views, only the new rows are written.
void GaussianModel::appendRows(const std::array<torch::Tensor, 5>& rows, int iteration)
{
	torch::NoGradGuard ng;
	const int64_t P = xyz_.size(0), n = rows[0].size(0);
	const bool have_state = groups_.size() == 5;
	bool live = arena_.capacity >= P + n && arena_.params[0][0][0].defined() && arena_.params[0][0][0].device() == xyz_.device();
	for (int i = 0; live && i < 5; i++) live = paramByIndex(i).data_ptr() == arena_.params[arena_.cur][i][0].data_ptr();
	// Lazy SH Adam: rows of the SH tensor may be steps behind.  An append IN PLACE moves no existing row, so they may stay
	// behind -- the new rows join up to date (zero moments: the zero-gradient steps they would take change nothing), and the
	// 1152 B per Gaussian of a flush (1 ms at 4 M Gaussians, measured: the mapper-loop leg of bench.py inserts every 10
	// iterations) are not paid.  Every other case re-seats the tensor: no row may be behind then.
	const bool keep_lazy = live && have_state && features_row_step_.defined();
	if (!keep_lazy) syncFeatures();
	if (!exist_since_iter_.defined()) exist_since_iter_ = torch::zeros({P}, xyz_.options().dtype(torch::kInt32).requires_grad(false));
	auto old_exist = exist_since_iter_;
	std::array<torch::Tensor, 5> old_param;
	std::array<std::array<torch::Tensor, 2>, 5> old_mom;
	for (int i = 0; i < 5; i++) {
		old_param[i] = paramByIndex(i).detach();
		if (have_state) old_mom[i] = {groups_[i].exp_avg, groups_[i].exp_avg_sq};
	}
	if (!live) {
		reserve(static_cast<int64_t>((P + n) * 1.25) + 64);   // (cur = 0; the old tensors stay alive through old_param / old_mom)
	}
	auto& cur = arena_.params[arena_.cur];
	for (int i = 0; i < 5; i++) {
		std::array<torch::Tensor, 3> b;
		for (int k = 0; k < 3; k++) b[k] = cur[i][k].narrow(0, 0, P + n);
		if (!live) {
			b[0].narrow(0, 0, P).copy_(old_param[i]);
			if (have_state) {
				b[1].narrow(0, 0, P).copy_(old_mom[i][0]);
				b[2].narrow(0, 0, P).copy_(old_mom[i][1]);
			}
		}
		b[0].narrow(0, P, n).copy_(rows[i].reshape(b[0].narrow(0, P, n).sizes()));
		b[1].narrow(0, P, n).zero_();
		b[2].narrow(0, P, n).zero_();
		if (have_state) replaceParam(i, b[0], b[1], b[2], /*rows_kept=*/keep_lazy);
		else replaceParam(i, b[0], torch::Tensor(), torch::Tensor());
	}
	if (keep_lazy)
		features_row_step_ = torch::cat({features_row_step_, torch::full({n}, groups_[1].step, features_row_step_.options())});
	auto exist = arena_.exist[arena_.cur].narrow(0, 0, P + n);
	if (old_exist.data_ptr() != exist.data_ptr()) exist.narrow(0, 0, P).copy_(old_exist);
	exist.narrow(0, P, n).fill_(iteration);
	exist_since_iter_ = exist;
	std::array<torch::Tensor, 3> stats;
	for (int k = 0; k < 3; k++) stats[k] = arena_.stats[arena_.cur][k].narrow(0, 0, P + n).zero_();   // :709-711
	xyz_gradient_accum_ = stats[0];
	denom_ = stats[1];
	max_radii2D_ = stats[2];
}

void GaussianModel::releaseArena()
{
	torch::NoGradGuard ng;
	syncFeatures();
	const bool have_state = groups_.size() == 5;
	for (int i = 0; i < 5; i++) {
		auto fresh = paramByIndex(i).detach().clone();
		if (have_state) replaceParam(i, fresh, groups_[i].exp_avg.clone(), groups_[i].exp_avg_sq.clone());
		else replaceParam(i, fresh, torch::Tensor(), torch::Tensor());
	}
	if (xyz_gradient_accum_.defined()) xyz_gradient_accum_ = xyz_gradient_accum_.clone();
	if (denom_.defined()) denom_ = denom_.clone();
	if (max_radii2D_.defined()) max_radii2D_ = max_radii2D_.clone();
	if (exist_since_iter_.defined()) exist_since_iter_ = exist_since_iter_.clone();
	arena_ = Arena();
	densify_scratch_ = torch::Tensor();
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
void GaussianModel::replaceParam(int group, torch::Tensor fresh, torch::Tensor exp_avg, torch::Tensor exp_avg_sq, bool rows_kept)
{
	// rows_kept: `fresh` is the old tensor's storage with rows appended -- the lazily stepped rows of the SH tensor stay where
	// (and as far behind as) they are; everything else re-seats the tensor and needs every row up to date first
	if (group == 1 && !rows_kept) syncFeatures();
	fresh = fresh.contiguous().set_requires_grad(true);
	paramByIndex(group) = fresh;
	if (static_cast<size_t>(group) < groups_.size()) {
		auto& g = groups_[static_cast<size_t>(group)];
		g.param = fresh;
		g.exp_avg = exp_avg.defined() ? exp_avg : torch::zeros_like(fresh);
		g.exp_avg_sq = exp_avg_sq.defined() ? exp_avg_sq : torch::zeros_like(fresh);
	}
}

// A leaf whose VALUES are replaced and whose moments start again from zero (resetOpacity, the loop-closure transforms: the
// reference's replaceTensorToOptimizer with a tensor of the same shape).  While the leaf is a view of the arena the new values
// are written into its rows and the moment rows are zeroed in place: the tensor stays where increasePcd / densifyAndPrune
// expect it.  (Re-seating it outside made the next increasePcd take the "not live" path: a fresh arena of 2 x 840 B per
// Gaussian and a copy of everything -- +25 ms on the iteration after a resetOpacity at 4 M Gaussians, bench.py --mapper-loop.)
void GaussianModel::replaceParamValues(int group, torch::Tensor fresh)
{
	auto& leaf = paramByIndex(group);
	const bool have_state = static_cast<size_t>(group) < groups_.size();
	bool in_arena = arena_.capacity > 0 && arena_.params[arena_.cur][group][0].defined() && leaf.defined() &&
	                leaf.data_ptr() == arena_.params[arena_.cur][group][0].data_ptr() && fresh.sizes() == leaf.sizes() &&
	                fresh.device() == leaf.device();
	if (!in_arena) {
		replaceParam(group, fresh, torch::Tensor(), torch::Tensor());
		return;
	}
	const int64_t P = leaf.size(0);
	auto& slot = arena_.params[arena_.cur][group];
	auto value = slot[0].narrow(0, 0, P);
	// (`fresh` may BE the leaf's rows -- a caller that edited them in place: nothing to copy, but the moment rows of the arena
	// are still zeroed below.  Installing zero moments OUTSIDE the arena there would leave the old ones in slot[1] / slot[2] for
	// the next in-place append to re-adopt, undoing the reset.)
	if (fresh.data_ptr() != value.data_ptr()) value.copy_(fresh.detach());
	if (have_state) replaceParam(group, value, slot[1].narrow(0, 0, P).zero_(), slot[2].narrow(0, 0, P).zero_());
	else replaceParam(group, value, torch::Tensor(), torch::Tensor());
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
	auto fresh = inverse_sigmoid(torch::min(act, bound)).detach();
	replaceParamValues(2, fresh);
}

// ---- loop closure (src/gaussian_model.cpp:379-475) -------------------------------------------------------------------------

void GaussianModel::applyScaledTransformation(const float s, torch::Tensor T)
{
	torch::NoGradGuard ng;
	// pt <- (s * Ryw * pt + tyw), :385-388: the positions are scaled in place, then transformed with the transposed matrix
	// (tensor_utils::EigenMatrix2TorchTensor(T.matrix()).transpose(0, 1): transformPoints reads it column-major)
	auto pts = (xyz_.detach() * s).contiguous();
	auto T_tensor = T.to(pts.device(), torch::kFloat32).transpose(0, 1);
	transformPoints(pts, T_tensor);
	// `this->scaling_ *= s` (:395): the log-scales multiplied by s, as shipped
	auto scl = (scaling_.detach() * s).contiguous();
	// scaledTransformationPostfix (:398-411): fresh leaves in groups 0 (xyz) and 4 (scaling; group 3 of the five here), zero
	// moments, the step counters stay
	replaceParamValues(0, pts);
	replaceParamValues(3, scl);
}

void GaussianModel::scaledTransformVisiblePointsOfKeyframe(torch::Tensor& point_not_transformed_flags, torch::Tensor& diff_pose,
                                                           torch::Tensor& kf_world_view_transform, torch::Tensor& kf_full_proj_transform,
                                                           const int kf_creation_iter, const int stable_num_iter_existence,
                                                           int& num_transformed, const float scale)
{
	torch::NoGradGuard ng;
	auto points = xyz_.detach().clone();                 // (the reference works on the leaf's storage and replaces the leaf anyway)
	auto rots = getRotationActivation().detach().clone();
	auto point_unstable_flags = torch::abs(exist_since_iter_ - kf_creation_iter) < stable_num_iter_existence;   // :433-436
	scaleAndTransformThenMarkVisiblePoints(points, rots, point_not_transformed_flags, point_unstable_flags, diff_pose,
	                                       kf_world_view_transform, kf_full_proj_transform, num_transformed, scale);
	replaceParamValues(0, points);   // :463  param_groups[0] = xyz_
	replaceParamValues(4, rots);     // :465  param_groups[5] = rotation_ (group 4 of the five here)
}

// ---- rebuilds as stream compaction (csrc/densify.hip, include/gsr.h) ------------------------------------------------
// gsr_densify_select turns the per-Gaussian decisions into a gather plan on the device, the host reads the counts ONCE (it
// has to size the new tensors), gsr_densify_gather rebuilds the five parameter tensors, their ten Adam moments and the
// statistics in one launch, out of the live tensors into the idle half of an arena.

// Two sets of [capacity, row] buffers for the parameters, their moments and the statistics: a growing map allocates
// nothing until it outgrows the capacity (then the arena grows by 1.5x).  Created on the first rebuild with 25 % headroom
// unless the caller reserved one.
void GaussianModel::reserve(int64_t capacity)
{
	const auto o = xyz_.options().requires_grad(false);
	const std::vector<std::vector<int64_t>> rows = {{3}, {features_.size(1), features_.size(2)}, {1}, {3}, {4}};
	for (int s = 0; s < 2; s++) {
		for (int i = 0; i < 5; i++) {
			std::vector<int64_t> shape = {capacity};
			shape.insert(shape.end(), rows[i].begin(), rows[i].end());
			for (int k = 0; k < 3; k++) arena_.params[s][i][k] = torch::empty(shape, o);
		}
		arena_.stats[s][0] = torch::empty({capacity, 1}, o);
		arena_.stats[s][1] = torch::empty({capacity, 1}, o);
		arena_.stats[s][2] = torch::empty({capacity}, o);
		arena_.exist[s] = torch::empty({capacity}, o.dtype(torch::kInt32));
	}
	arena_.capacity = capacity;
	arena_.cur = 0;
}

namespace {
void check_gsr(int st, const char* where)
{
	if (st != GSR_OK) throw std::runtime_error(std::string(where) + ": " + gsr_strerror(st) + " (" + gsr_last_hip_error_string() + ")");
}
}  // namespace

void* GaussianModel::hostStream(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}

std::array<int64_t, 6> GaussianModel::compact(gsr_densify_select_args& sel, c10::optional<at::Generator> generator, bool morton_reindex)
{
	torch::NoGradGuard ng;
	syncFeatures();   // the gather copies rows of features_ and of its moments: none may be behind (lazy SH Adam)
	const int64_t P = xyz_.size(0);
	sel.P = static_cast<int>(P);
	void* stream = hostStream(xyz_);
	const auto bytes = static_cast<int64_t>(gsr_densify_scratch_bytes(sel.P));
	if (!densify_scratch_.defined() || densify_scratch_.numel() < bytes || densify_scratch_.device() != xyz_.device())
		densify_scratch_ = torch::empty({bytes + bytes / 4 + 256}, xyz_.options().dtype(torch::kUInt8).requires_grad(false));
	auto counts = torch::empty({8}, xyz_.options().dtype(torch::kInt32).requires_grad(false));
	check_gsr(gsr_densify_select(&sel, reinterpret_cast<char*>(densify_scratch_.data_ptr<uint8_t>()), counts.data_ptr<int>(), stream),
	          "gsr_densify_select");
	auto host = counts.cpu();   // the one host read of the rebuild
	const int* c = host.data_ptr<int>();
	const int64_t n_keep = c[0], n_clone = c[1], n_child = c[2], n_split = c[3], n_clone_sel = c[4], n_new = c[5];
	if (arena_.capacity < n_new || !arena_.params[0][0][0].defined() || arena_.params[0][0][0].device() != xyz_.device())
		reserve(static_cast<int64_t>(std::max(n_new, P) * (arena_.capacity ? 1.5 : 1.25)) + 64);
	arena_.cur ^= 1;
	auto& dst = arena_.params[arena_.cur];
	// at::normal(zeros, stds) of the reference (:731-734) is randn(2k,3) * stds: the same draws, the scale applied in-kernel
	torch::Tensor samples;
	if (n_split) samples = torch::empty({2 * n_split, 3}, xyz_.options().requires_grad(false)).normal_(0.0, 1.0, generator);
	gsr_densify_gather_args g{};
	g.P = sel.P;
	g.n_new = (int)n_new; g.n_keep = (int)n_keep; g.n_clone = (int)n_clone; g.n_child = (int)n_child; g.n_split = (int)n_split;
	g.features_row_floats = static_cast<int>(features_.size(1) * features_.size(2));
	std::vector<torch::Tensor> keep_alive;
	const bool have_state = groups_.size() == 5;
	std::array<std::array<torch::Tensor, 3>, 5> out;
	for (int i = 0; i < 5; i++) {
		auto src = paramByIndex(i).detach().contiguous();
		keep_alive.push_back(src);
		for (int k = 0; k < 3; k++) out[i][k] = dst[i][k].narrow(0, 0, n_new);
		g.param_in[i] = src.data_ptr<float>();
		g.param_out[i] = out[i][0].data_ptr<float>();
		if (have_state) {
			g.exp_avg_in[i] = groups_[i].exp_avg.data_ptr<float>();
			g.exp_avg_sq_in[i] = groups_[i].exp_avg_sq.data_ptr<float>();
			g.exp_avg_out[i] = out[i][1].data_ptr<float>();
			g.exp_avg_sq_out[i] = out[i][2].data_ptr<float>();
		}
	}
	g.samples = samples.defined() ? samples.data_ptr<float>() : nullptr;
	std::array<torch::Tensor, 3> stats;
	for (int k = 0; k < 3; k++) {
		stats[k] = arena_.stats[arena_.cur][k].narrow(0, 0, n_new);
		g.stats_out[k] = stats[k].data_ptr<float>();
	}
	// exist_since_iter_: every row of the new set inherits its source's value (:636, :744, :782)
	torch::Tensor exist_old, exist_new;
	if (exist_since_iter_.defined() && exist_since_iter_.numel() == P) {
		exist_old = exist_since_iter_.contiguous();
		exist_new = arena_.exist[arena_.cur].narrow(0, 0, n_new);
		g.exist_since_iter_in = exist_old.data_ptr<int>();
		g.exist_since_iter_out = exist_new.data_ptr<int>();
	}
	torch::Tensor morton;
	if (morton_reindex && n_new) {   // (densifyAndPrune only: prunePoints copies its statistics by the mask's order)
		morton = torch::empty({static_cast<int64_t>(gsr_densify_morton_scratch_bytes((int)n_new)) + 256}, xyz_.options().dtype(torch::kUInt8).requires_grad(false));
		g.morton_scratch = reinterpret_cast<char*>(morton.data_ptr<uint8_t>());
	}
	if (n_new) check_gsr(gsr_densify_gather(&g, reinterpret_cast<const char*>(densify_scratch_.data_ptr<uint8_t>()), stream), "gsr_densify_gather");
	if (exist_new.defined()) exist_since_iter_ = exist_new;
	for (int i = 0; i < 5; i++) {
		if (have_state) replaceParam(i, out[i][0], out[i][1], out[i][2]);
		else replaceParam(i, out[i][0], torch::Tensor(), torch::Tensor());
	}
	xyz_gradient_accum_ = stats[0];
	denom_ = stats[1];
	max_radii2D_ = stats[2];
	return {n_keep, n_clone, n_child, n_split, n_clone_sel, n_new};
}

// src/gaussian_model.cpp:588-642 (the statistics keep their values here, unlike in densifyAndPrune)
void GaussianModel::prunePoints(torch::Tensor& mask)
{
	torch::NoGradGuard ng;
	auto m8 = mask.to(torch::kUInt8).contiguous();
	auto old_accum = xyz_gradient_accum_, old_denom = denom_, old_max = max_radii2D_;
	gsr_densify_select_args sel{};
	sel.prune_mask = m8.data_ptr<uint8_t>();
	compact(sel, c10::nullopt);
	auto keep = ~mask.to(torch::kBool);
	xyz_gradient_accum_.copy_(old_accum.index({keep}));
	denom_.copy_(old_denom.index({keep}));
	max_radii2D_.copy_(old_max.index({keep}));
}

// The whole model -- parameters, Adam moments, statistics, exist_since_iter_ -- laid out along a Z-order curve of the positions (what
// morton_reindex_ does inside densifyAndPrune, on request: a map that no longer densifies keeps growing at its end through
// increasePcd).  The same Gaussians with the same values; returns perm with new row r = old row perm[r].  The permutation is read
// off the gather itself: exist_since_iter_ travels through it as the row number.
torch::Tensor GaussianModel::reorderAlongZCurve()
{
	torch::NoGradGuard ng;
	const int64_t P = xyz_.size(0);
	const auto iopt = xyz_.options().dtype(torch::kInt32).requires_grad(false);
	if (P == 0) return torch::empty({0}, iopt.dtype(torch::kInt64));
	auto old_accum = xyz_gradient_accum_, old_denom = denom_, old_max = max_radii2D_, old_exist = exist_since_iter_;
	const bool had_exist = old_exist.defined() && old_exist.numel() == P;
	exist_since_iter_ = torch::arange(P, iopt);
	auto none = torch::zeros({P}, iopt.dtype(torch::kUInt8));
	gsr_densify_select_args sel{};
	sel.prune_mask = none.data_ptr<uint8_t>();
	compact(sel, c10::nullopt, /*morton_reindex=*/true);
	auto perm = exist_since_iter_.to(torch::kInt64);   // (a copy: exist_since_iter_ is a view into the arena)
	xyz_gradient_accum_.copy_(old_accum.index({perm}));
	denom_.copy_(old_denom.index({perm}));
	max_radii2D_.copy_(old_max.index({perm}));
	if (had_exist) exist_since_iter_.copy_(old_exist.index({perm}));
	else exist_since_iter_ = torch::Tensor();
	return perm;
}

// src/gaussian_model.cpp:795-815 with densifyAndClone (:763-793), densifyAndSplit (:716-761, N = 2) and the final
// prunePoints (:805-813) folded into one rebuild.
GaussianModel::DensifyResult GaussianModel::densifyAndPrune(float max_grad, float min_opacity, float extent,
                                                            int max_screen_size, c10::optional<at::Generator> generator)
{
	torch::NoGradGuard ng;
	const int64_t P = xyz_.size(0);
	auto accum = xyz_gradient_accum_.contiguous(), denom = denom_.contiguous();
	auto scaling = scaling_.detach().contiguous(), opacity = opacity_.detach().contiguous();
	gsr_densify_select_args sel{};
	sel.xyz_gradient_accum = accum.data_ptr<float>();
	sel.denom = denom.data_ptr<float>();
	sel.scaling = scaling.data_ptr<float>();
	sel.opacity = opacity.data_ptr<float>();
	sel.percent_dense = opt_.percent_dense_;
	sel.max_grad = max_grad;
	sel.min_opacity = min_opacity;
	sel.extent = extent;
	sel.max_screen_size = max_screen_size;
	const auto c = compact(sel, generator, morton_reindex_);
	DensifyResult res;
	res.cloned = c[4];
	res.split = c[3];
	res.pruned = (P - c[3] - c[0]) + (c[4] - c[1]) + 2 * (c[3] - c[2]);
	res.points = c[5];
	return res;
}
