// ops_register.cpp -- exposes the C++ host layer to Python (torch.ops.photoslam_amd.*) so that the
// test-suite and bench.py can drive the very code a C++ caller (gaussian_mapper) would link against.
#include <torch/library.h>
#include "loss_utils.h"
#include <torch/torch.h>

#include <map>
#include <mutex>

#include "gaussian_model_lite.h"
#include "gaussian_renderer.h"
#include "keyframe_scheduler.h"
#include <torch/csrc/distributed/c10d/GroupRegistry.hpp>
#include "operate_points.h"
#include "spatial.h"
#include "stereo_vision.h"

namespace {

std::tuple<torch::Tensor, torch::Tensor> rasterize_gaussians(torch::Tensor means3D, torch::Tensor means2D, torch::Tensor sh,
                                                             torch::Tensor colors_precomp, torch::Tensor opacities,
                                                             torch::Tensor scales, torch::Tensor rotations,
                                                             torch::Tensor cov3Ds_precomp, torch::Tensor bg,
                                                             double scale_modifier, torch::Tensor viewmatrix,
                                                             torch::Tensor projmatrix, double tanfovx, double tanfovy,
                                                             int64_t image_height, int64_t image_width, int64_t sh_degree,
                                                             torch::Tensor campos, bool prefiltered)
{
	GaussianRasterizationSettings s((int)image_height, (int)image_width, (float)tanfovx, (float)tanfovy, bg,
	                                (float)scale_modifier, viewmatrix, projmatrix, (int)sh_degree, campos, prefiltered);
	GaussianRasterizer r(s);
	auto has = [](const torch::Tensor& t) { return t.defined() && t.numel() != 0; };
	return r.forward(means3D, means2D, opacities, has(sh), has(colors_precomp), has(scales), has(rotations),
	                 has(cov3Ds_precomp), sh, colors_precomp, scales, rotations, cov3Ds_precomp);
}

torch::Tensor mark_visible(torch::Tensor means3D, torch::Tensor viewmatrix, torch::Tensor projmatrix)
{
	return markVisible(means3D, viewmatrix, projmatrix);
}
torch::Tensor dist_cuda2(torch::Tensor points) { return distCUDA2(points); }
torch::Tensor l1_ssim_loss(torch::Tensor rendered, torch::Tensor gt, torch::Tensor mask, double lambda_dssim)
{
	return fusedL1SSIMLoss(rendered, gt, mask, (float)lambda_dssim);
}

torch::Tensor transform_points(torch::Tensor points, torch::Tensor transformmatrix)
{
	transformPoints(points, transformmatrix);
	return points;
}
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, int64_t> scale_transform_mark_visible(
    torch::Tensor points, torch::Tensor rots, torch::Tensor not_transformed, torch::Tensor unstable, torch::Tensor transform,
    torch::Tensor view, torch::Tensor proj, int64_t num_transformed, double scale)
{
	int n = (int)num_transformed;
	scaleAndTransformThenMarkVisiblePoints(points, rots, not_transformed, unstable, transform, view, proj, n, (float)scale);
	return std::make_tuple(points, rots, not_transformed, (int64_t)n);
}
torch::Tensor reproject_depth_pinhole(torch::Tensor depth, torch::Tensor mask, std::vector<double> intr, int64_t width)
{
	std::vector<float> f(intr.begin(), intr.end());
	return reprojectDepthPinhole(depth, mask, f, (int)width);
}
std::tuple<torch::Tensor, torch::Tensor> neighborhood_keypoints(torch::Tensor px, torch::Tensor has, torch::Tensor p3,
                                                                torch::Tensor colors, double max_dist, std::vector<double> intr,
                                                                int64_t width)
{
	std::vector<float> f(intr.begin(), intr.end());
	return monocularPinholeInactiveGeoDensifyBySearchingNeighborhoodKeypoints(px, has, p3, colors, (float)max_dist, f, (int)width);
}

// ---- a C++ TrainStep behind an integer handle
std::mutex g_mu;
std::map<int64_t, std::shared_ptr<TrainStep>> g_trainers;
int64_t g_next = 1;

int64_t trainer_create(torch::Tensor xyz, torch::Tensor features, torch::Tensor opacity, torch::Tensor scaling,
                       torch::Tensor rotation, int64_t sh_degree, double spatial_lr_scale, torch::Tensor background)
{
	auto model = std::make_shared<GaussianModel>((int)sh_degree, xyz, features, opacity, scaling, rotation,
	                                             (float)spatial_lr_scale);
	model->trainingSetup(GaussianOptimizationParams());
	std::lock_guard<std::mutex> lk(g_mu);
	g_trainers[g_next] = std::make_shared<TrainStep>(model, background);
	return g_next++;
}
int64_t trainer_create_from_pcd(torch::Tensor points, torch::Tensor colors, int64_t sh_degree, double spatial_lr_scale,
                                torch::Tensor background)
{
	auto model = std::make_shared<GaussianModel>((int)sh_degree);
	model->createFromPcd(points, colors, (float)spatial_lr_scale);
	model->trainingSetup(GaussianOptimizationParams());
	std::lock_guard<std::mutex> lk(g_mu);
	g_trainers[g_next] = std::make_shared<TrainStep>(model, background);
	return g_next++;
}
// a generator of the tensors' device seeded like torch.Generator(device).manual_seed(seed)
at::Generator make_generator(const torch::Device& device, int64_t seed)
{
	auto gen = at::globalContext().defaultGenerator(device).clone();
	gen.set_current_seed(static_cast<uint64_t>(seed));
	return gen;
}
std::shared_ptr<TrainStep> get(int64_t h)
{
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_trainers.find(h);
	TORCH_CHECK(it != g_trainers.end(), "unknown trainer handle");
	return it->second;
}
std::shared_ptr<GaussianKeyframe> make_kf(torch::Tensor view, torch::Tensor proj, torch::Tensor campos, double fovx,
                                          double fovy, int64_t H, int64_t W)
{
	auto kf = std::make_shared<GaussianKeyframe>();
	kf->image_height_ = (int)H;
	kf->image_width_ = (int)W;
	kf->FoVx_ = (float)fovx;
	kf->FoVy_ = (float)fovy;
	kf->world_view_transform_ = view;
	kf->full_proj_transform_ = proj;
	kf->camera_center_ = campos;
	return kf;
}
torch::Tensor trainer_render_and_backward(int64_t h, torch::Tensor view, torch::Tensor proj, torch::Tensor campos,
                                          double fovx, double fovy, int64_t H, int64_t W, torch::Tensor gt,
                                          torch::Tensor mask)
{
	return get(h)->renderAndBackward(make_kf(view, proj, campos, fovx, fovy, H, W), gt, mask).detach();
}
// GaussianRenderer::render on the trainer's model with the pipeline flags of GaussianPipelineParams (convert_SHs_,
// compute_cov3D_: src/gaussian_renderer.cpp:78-113): (image, radii); the image is attached to the model's leaves
std::tuple<torch::Tensor, torch::Tensor> trainer_render(int64_t h, torch::Tensor view, torch::Tensor proj, torch::Tensor campos,
                                                        double fovx, double fovy, int64_t H, int64_t W, bool convert_SHs,
                                                        bool compute_cov3D, bool fuse_activations)
{
	auto t = get(h);
	GaussianPipelineParams pipe;
	pipe.convert_SHs_ = convert_SHs;
	pipe.compute_cov3D_ = compute_cov3D;
	torch::Tensor override_color;
	auto pkg = GaussianRenderer::render(make_kf(view, proj, campos, fovx, fovy, H, W), (int)H, (int)W, t->gaussians_, pipe,
	                                    t->background_, override_color, 1.0f, false, fuse_activations);
	return std::make_tuple(std::get<0>(pkg), std::get<3>(pkg));
}
void trainer_finish(int64_t h) { get(h)->finishOneIteration(); }
void trainer_finish_begin(int64_t h) { get(h)->finishBegin(); }
void trainer_adam_group(int64_t h, int64_t group) { get(h)->finishAdamGroup(static_cast<int>(group)); }
void trainer_finish_end(int64_t h) { get(h)->finishEnd(); }
std::vector<torch::Tensor> trainer_params(int64_t h) { return get(h)->gaussians_->params(); }
std::vector<torch::Tensor> trainer_grads(int64_t h)
{
	std::vector<torch::Tensor> g;
	// (features_ has no gradient yet in the factored mode: an empty tensor keeps the positions)
	for (auto& p : get(h)->gaussians_->paramsRaw()) g.push_back(p.grad().defined() ? p.grad() : torch::empty({0}, p.options()));
	return g;
}
// densification: options of the schedule (names of GaussianOptimizationParams / GaussianMapper without the trailing
// underscore; unknown keys are refused), and the single operations for the parity tests
void trainer_set_options(int64_t h, c10::Dict<std::string, double> o)
{
	auto t = get(h);
	auto& p = t->gaussians_->opt_;
	for (const auto& kv : o) {
		const std::string& k = kv.key();
		const double v = kv.value();
		if (k == "densify") t->densify_ = v != 0.0;
		else if (k == "fused_sh_adam") t->fused_sh_adam_ = v != 0.0;
		else if (k == "lazy_sh_adam_window") t->lazy_sh_adam_window_ = (int)v;
		else if (k == "fused_geom_adam") t->fused_geom_adam_ = v != 0.0;
		else if (k == "active_sh_degree") t->gaussians_->active_sh_degree_ = std::min((int)v, t->gaussians_->max_sh_degree_);
		else if (k == "cull_empty_tiles") t->cull_empty_tiles_ = v != 0.0;
		else if (k == "persistent_workspace") t->persistent_workspace_ = v != 0.0;
		else if (k == "early_gather") t->early_gather_ = v != 0.0;
		else if (k == "packed_exchange") t->packed_exchange_ = v != 0.0;
		else if (k == "pack_in_backward") t->pack_in_backward_ = v != 0.0;
		else if (k == "lazy_slice_late") t->lazy_slice_late_ = v != 0.0;
		else if (k == "no_side_stream") t->no_side_stream_ = v != 0.0;
		else if (k == "profile_exchange") t->profile_exchange_ = v != 0.0;
		else if (k == "convert_SHs") t->pipe_.convert_SHs_ = v != 0.0;
		else if (k == "compute_cov3D") t->pipe_.compute_cov3D_ = v != 0.0;
		else if (k == "cameras_extent") t->cameras_extent_ = (float)v;
		else if (k == "position_lr_step") t->position_lr_step_ = (int)v;
		else if (k == "morton_reindex") t->gaussians_->morton_reindex_ = v != 0.0;
		else if (k == "lr_scale") t->gaussians_->lr_scale_ = v;
		else if (k == "densify_min_opacity") t->densify_min_opacity_ = (float)v;
		else if (k == "prune_big_point_after_iter") t->prune_big_point_after_iter_ = (int)v;
		else if (k == "seed") t->generator_ = make_generator(t->gaussians_->xyz_.device(), (int64_t)v);
		else if (k == "iterations") p.iterations_ = (int)v;
		else if (k == "densify_from_iter") p.densify_from_iter_ = (int)v;
		else if (k == "densify_until_iter") p.densify_until_iter_ = (int)v;
		else if (k == "densification_interval") p.densification_interval_ = (int)v;
		else if (k == "opacity_reset_interval") p.opacity_reset_interval_ = (int)v;
		else if (k == "densify_grad_threshold") p.densify_grad_threshold_ = (float)v;
		else if (k == "percent_dense") p.percent_dense_ = (float)v;
		else TORCH_CHECK(false, "unknown trainer option: ", k);
	}
}
std::vector<int64_t> trainer_densify_and_prune(int64_t h, double max_grad, double min_opacity, double extent,
                                               int64_t max_screen_size, int64_t seed)
{
	auto g = get(h)->gaussians_;
	auto r = g->densifyAndPrune((float)max_grad, (float)min_opacity, (float)extent, (int)max_screen_size,
	                            make_generator(g->xyz_.device(), seed));
	return {r.cloned, r.split, r.pruned, r.points};
}
torch::Tensor trainer_reorder_along_z_curve(int64_t h) { return get(h)->gaussians_->reorderAlongZCurve(); }
std::vector<int64_t> trainer_last_densify(int64_t h)
{
	auto r = get(h)->last_densify_;
	return {r.cloned, r.split, r.pruned, r.points};
}
void trainer_save_ply(int64_t h, std::string path) { get(h)->gaussians_->savePly(path); }
// a fresh trainer from a PLY checkpoint (GaussianModel::loadPly): the tensors land on the device of `bg`
int64_t trainer_create_from_ply(std::string path, int64_t sh_degree, double spatial_lr_scale, torch::Tensor bg)
{
	auto g = std::make_shared<GaussianModel>((int)sh_degree);
	g->device_ = bg.device();
	g->loadPly(path);
	g->spatial_lr_scale_ = (float)spatial_lr_scale;
	g->trainingSetup(GaussianOptimizationParams());
	std::lock_guard<std::mutex> lk(g_mu);
	g_trainers[g_next] = std::make_shared<TrainStep>(g, bg);
	return g_next++;
}
void trainer_reset_opacity(int64_t h) { get(h)->gaussians_->resetOpacity(); }
void trainer_apply_scaled_transformation(int64_t h, double s, torch::Tensor T) { get(h)->gaussians_->applyScaledTransformation((float)s, T); }
// -> (flags after the call, number of points moved)
std::tuple<torch::Tensor, int64_t> trainer_scaled_transform_visible(int64_t h, torch::Tensor flags, torch::Tensor diff_pose,
                                                                    torch::Tensor view, torch::Tensor proj, int64_t kf_creation_iter,
                                                                    int64_t stable_num_iter_existence, double scale)
{
	int moved = 0;
	auto f = flags.clone();
	get(h)->gaussians_->scaledTransformVisiblePointsOfKeyframe(f, diff_pose, view, proj, (int)kf_creation_iter, (int)stable_num_iter_existence,
	                                                          moved, (float)scale);
	return {f, (int64_t)moved};
}
void trainer_prune_points(int64_t h, torch::Tensor mask) { get(h)->gaussians_->prunePoints(mask); }
void trainer_one_up_sh_degree(int64_t h) { get(h)->gaussians_->oneUpShDegree(); }
std::vector<torch::Tensor> trainer_moments(int64_t h)   // exp_avg of the five groups, then exp_avg_sq
{
	std::vector<torch::Tensor> out;
	get(h)->gaussians_->syncFeatures();
	for (auto& g : get(h)->gaussians_->groups_) out.push_back(g.exp_avg);
	for (auto& g : get(h)->gaussians_->groups_) out.push_back(g.exp_avg_sq);
	return out;
}

// lazy SH Adam: the step every row of the SH tensor has taken (empty = every row is up to date), WITHOUT bringing them up to date
torch::Tensor trainer_features_row_step(int64_t h)
{
	auto& r = get(h)->gaussians_->features_row_step_;
	return r.defined() ? r.clone() : torch::empty({0}, torch::kInt32);
}

// Adam step counters of the five groups (torch::optim::AdamParamState::step of the reference's six: features_dc and
// features_rest share one counter here, as they always carry a gradient together)
std::vector<int64_t> trainer_steps(int64_t h)
{
	std::vector<int64_t> out;
	for (auto& g : get(h)->gaussians_->groups_) out.push_back(g.step);
	return out;
}
void trainer_set_steps(int64_t h, std::vector<int64_t> steps)
{
	auto& groups = get(h)->gaussians_->groups_;
	TORCH_CHECK(steps.size() == groups.size(), "one step counter per parameter group");
	for (size_t i = 0; i < groups.size(); i++) groups[i].step = (int)steps[i];
}

bool trainer_densify_due(int64_t h) { return get(h)->densifyDue(); }

// view-factored exchange of the data-parallel step (bench.py --gpus N, trainer.ViewFactoredExchange)
void trainer_set_factored_exchange(int64_t h, bool on) { get(h)->factored_exchange_ = on; }
torch::Tensor trainer_sh_grad_view(int64_t h) { return get(h)->sh_grad_view_; }
torch::Tensor trainer_sh_send_buffer(int64_t h) { return get(h)->sh_send_; }
std::vector<double> trainer_exchange_wait_ms(int64_t h) { return get(h)->exchangeWaitMs(); }
void trainer_features_grad_from_views(int64_t h, torch::Tensor campos_views, torch::Tensor views)
{
	get(h)->setFeaturesGradFromViews(campos_views, views);
}
void trainer_features_step_from_views(int64_t h, torch::Tensor campos_views, torch::Tensor views, int64_t row0, bool first_part)
{
	get(h)->stepFeaturesFromViews(campos_views, views, row0, first_part);
}
// GaussianModel::increasePcd (src/gaussian_model.cpp:188-376): the tensor overload, or -- vector_overload -- the std::vector one
void trainer_increase_pcd(int64_t h, torch::Tensor points, torch::Tensor colors, int64_t iteration, bool vector_overload)
{
	auto g = get(h)->gaussians_;
	if (vector_overload) {
		auto p = points.detach().to(torch::kCPU).to(torch::kFloat32).contiguous(), c = colors.detach().to(torch::kCPU).to(torch::kFloat32).contiguous();
		std::vector<float> pv(p.data_ptr<float>(), p.data_ptr<float>() + p.numel()), cv(c.data_ptr<float>(), c.data_ptr<float>() + c.numel());
		g->increasePcd(pv, cv, (int)iteration);
	} else {
		g->increasePcd(points, colors, (int)iteration);
	}
}
torch::Tensor trainer_exist_since_iter(int64_t h) { return get(h)->gaussians_->exist_since_iter_; }
void trainer_set_exist_since_iter(int64_t h, torch::Tensor v)
{
	auto g = get(h)->gaussians_;
	g->exist_since_iter_ = v.to(g->xyz_.device()).to(torch::kInt32).contiguous().clone();
}
void trainer_release_arena(int64_t h) { get(h)->gaussians_->releaseArena(); }
// Data-parallel keyframe batches driven from C++: the process group Python created (torch.distributed's default group, or any
// other) is resolved by its registered name; from then on trainer_train_one_iteration() issues every collective itself.
// the host-side group (gloo) over which the packed exchange agrees on its message capacity; empty name = none
void trainer_set_count_group(int64_t h, std::string group_name)
{
	if (group_name.empty()) get(h)->setCountGroup(nullptr);
	else get(h)->setCountGroup(c10d::resolve_process_group(group_name));
}
void trainer_set_process_group(int64_t h, std::string group_name, bool factored)
{
	if (group_name.empty()) get(h)->setProcessGroup(nullptr, factored);
	else get(h)->setProcessGroup(c10d::resolve_process_group(group_name), factored);
}
torch::Tensor trainer_train_one_iteration(int64_t h, torch::Tensor view, torch::Tensor proj, torch::Tensor campos, double fovx, double fovy,
                                          int64_t height, int64_t width, torch::Tensor gt, torch::Tensor mask)
{
	return get(h)->trainForOneIteration(make_kf(view, proj, campos, fovx, fovy, height, width), gt, mask).detach();
}
void trainer_features_finish_from_views(int64_t h) { get(h)->finishFeaturesFromViews(); }
void trainer_geom_adam(int64_t h, double grad_scale) { get(h)->finishGeomAdam((float)grad_scale); }
torch::Tensor sh_grad_from_views(torch::Tensor means3D, torch::Tensor campos_views, torch::Tensor views, int64_t degree,
                                 int64_t M, double scale)
{
	return shGradFromViews(means3D, campos_views, views, (int)degree, (int)M, (float)scale);
}
std::vector<torch::Tensor> trainer_stats(int64_t h)
{
	auto g = get(h)->gaussians_;
	return {g->xyz_gradient_accum_, g->denom_, g->max_radii2D_};
}
// KeyframeScheduler (host/include/keyframe_scheduler.h) behind handles, for the tests and for a Python driver of a keyframe batch
std::map<int64_t, std::shared_ptr<KeyframeScheduler>> g_schedulers;
int64_t g_next_scheduler = 1;
std::shared_ptr<KeyframeScheduler> scheduler(int64_t h)
{
	std::lock_guard<std::mutex> lk(g_mu);
	auto it = g_schedulers.find(h);
	TORCH_CHECK(it != g_schedulers.end(), "unknown keyframe scheduler handle ", h);
	return it->second;
}
int64_t keyframe_scheduler_create(int64_t seed)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_schedulers[g_next_scheduler] = std::make_shared<KeyframeScheduler>((uint64_t)seed);
	return g_next_scheduler++;
}
void keyframe_scheduler_destroy(int64_t h)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_schedulers.erase(h);
}
int64_t keyframe_scheduler_add(int64_t h, int64_t times_of_use) { return scheduler(h)->addKeyframe((int)times_of_use); }
void keyframe_scheduler_increase(int64_t h, int64_t keyframe, int64_t times) { scheduler(h)->increaseTimesOfUse((int)keyframe, (int)times); }
int64_t keyframe_scheduler_use_one(int64_t h) { return scheduler(h)->useOne(); }
std::vector<int64_t> keyframe_scheduler_use_batch(int64_t h, int64_t B)
{
	const auto b = scheduler(h)->useBatch((int)B);
	return std::vector<int64_t>(b.begin(), b.end());
}
std::vector<int64_t> keyframe_scheduler_use_batch_on_ranks(int64_t h, std::string group_name)
{
	const auto b = scheduler(h)->useBatchOnRanks(c10d::resolve_process_group(group_name));
	return std::vector<int64_t>(b.begin(), b.end());
}
// [used times, remaining times of use] of every keyframe
std::vector<int64_t> keyframe_scheduler_state(int64_t h)
{
	auto s = scheduler(h);
	std::vector<int64_t> out;
	for (int k = 0; k < s->size(); k++) out.push_back(s->usedTimes(k));
	for (int k = 0; k < s->size(); k++) out.push_back(s->remainingTimesOfUse(k));
	return out;
}
void trainer_destroy(int64_t h)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_trainers.erase(h);
}

}  // namespace

TORCH_LIBRARY(photoslam_amd, m)
{
	m.def("rasterize_gaussians", &rasterize_gaussians);
	m.def("mark_visible", &mark_visible);
	m.def("dist_cuda2", &dist_cuda2);
	m.def("l1_ssim_loss", &l1_ssim_loss);
	// host/include/loss_utils.h, the functions behind the reference's names (tests pin them to the reference's header compiled)
	m.def("loss_ssim", +[](torch::Tensor a, torch::Tensor b, int64_t window_size, bool size_average) {
		return loss_utils::ssim(a, b, a.device().type(), (int)window_size, size_average);
	});
	m.def("loss_ssim_with_window", +[](torch::Tensor a, torch::Tensor b, torch::Tensor window, int64_t window_size, bool size_average) {
		torch::autograd::Variable w = window;
		return loss_utils::_ssim(a, b, w, (int)window_size, a.size(-3), size_average);
	});
	m.def("loss_psnr", +[](torch::Tensor a, torch::Tensor b) { return loss_utils::psnr(a, b); });
	m.def("loss_psnr_gaussian_splatting", +[](torch::Tensor a, torch::Tensor b) { return loss_utils::psnr_gaussian_splatting(a, b); });
	m.def("loss_create_window", +[](int64_t window_size, int64_t channel, torch::Tensor like) {
		return loss_utils::create_window((int)window_size, channel, like.device().type());
	});
	m.def("loss_l1", +[](torch::Tensor a, torch::Tensor b) { return loss_utils::l1_loss(a, b); });
	m.def("transform_points", &transform_points);
	m.def("scale_transform_mark_visible", &scale_transform_mark_visible);
	m.def("reproject_depth_pinhole", &reproject_depth_pinhole);
	m.def("neighborhood_keypoints", &neighborhood_keypoints);
	m.def("trainer_create", &trainer_create);
	m.def("trainer_render_and_backward", &trainer_render_and_backward);
	m.def("trainer_render", &trainer_render);
	m.def("trainer_finish", &trainer_finish);
	m.def("trainer_finish_begin", &trainer_finish_begin);
	m.def("trainer_adam_group", &trainer_adam_group);
	m.def("trainer_finish_end", &trainer_finish_end);
	m.def("trainer_params", &trainer_params);
	m.def("trainer_grads", &trainer_grads);
	m.def("trainer_stats", &trainer_stats);
	m.def("trainer_create_from_pcd", &trainer_create_from_pcd);
	m.def("trainer_set_options", &trainer_set_options);
	m.def("trainer_densify_and_prune", &trainer_densify_and_prune);
	m.def("trainer_reorder_along_z_curve", &trainer_reorder_along_z_curve);
	m.def("trainer_last_densify", &trainer_last_densify);
	m.def("trainer_save_ply", &trainer_save_ply);
	m.def("trainer_create_from_ply", &trainer_create_from_ply);
	m.def("trainer_reset_opacity", &trainer_reset_opacity);
	m.def("trainer_apply_scaled_transformation", &trainer_apply_scaled_transformation);
	m.def("trainer_scaled_transform_visible", &trainer_scaled_transform_visible);
	m.def("trainer_prune_points", &trainer_prune_points);
	m.def("trainer_one_up_sh_degree", &trainer_one_up_sh_degree);
	m.def("trainer_moments", &trainer_moments);
	m.def("trainer_steps", &trainer_steps);
	m.def("trainer_features_row_step", &trainer_features_row_step);
	m.def("trainer_set_steps", &trainer_set_steps);
	m.def("trainer_densify_due", &trainer_densify_due);
	m.def("trainer_set_factored_exchange", &trainer_set_factored_exchange);
	m.def("trainer_sh_grad_view", &trainer_sh_grad_view);
	m.def("trainer_sh_send_buffer", &trainer_sh_send_buffer);
	m.def("trainer_exchange_wait_ms", &trainer_exchange_wait_ms);
	m.def("trainer_features_grad_from_views", &trainer_features_grad_from_views);
	m.def("trainer_features_step_from_views", &trainer_features_step_from_views);
	m.def("trainer_increase_pcd", &trainer_increase_pcd);
	m.def("trainer_exist_since_iter", &trainer_exist_since_iter);
	m.def("trainer_set_exist_since_iter", &trainer_set_exist_since_iter);
	m.def("trainer_release_arena", &trainer_release_arena);
	m.def("trainer_set_process_group", &trainer_set_process_group);
	m.def("trainer_set_count_group", &trainer_set_count_group);
	m.def("trainer_train_one_iteration", &trainer_train_one_iteration);
	m.def("trainer_features_finish_from_views", &trainer_features_finish_from_views);
	m.def("trainer_geom_adam", &trainer_geom_adam);
	m.def("sh_grad_from_views", &sh_grad_from_views);
	m.def("trainer_destroy", &trainer_destroy);
	m.def("keyframe_scheduler_create", &keyframe_scheduler_create);
	m.def("keyframe_scheduler_destroy", &keyframe_scheduler_destroy);
	m.def("keyframe_scheduler_add", &keyframe_scheduler_add);
	m.def("keyframe_scheduler_increase", &keyframe_scheduler_increase);
	m.def("keyframe_scheduler_use_one", &keyframe_scheduler_use_one);
	m.def("keyframe_scheduler_use_batch", &keyframe_scheduler_use_batch);
	m.def("keyframe_scheduler_use_batch_on_ranks", &keyframe_scheduler_use_batch_on_ranks);
	m.def("keyframe_scheduler_state", &keyframe_scheduler_state);
}
