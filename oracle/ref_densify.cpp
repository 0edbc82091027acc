/*
 * ref_densify.cpp -- torch ops around the REFERENCE's own map-maintenance code: the member functions
 *   GaussianModel::{getXYZ, getScalingActivation, getOpacityActivation, trainingSetup, resetOpacity,
 *   replaceTensorToOptimizer, prunePoints, densificationPostfix, densifyAndSplit, densifyAndClone, densifyAndPrune,
 *   addDensificationStats, percentDense, setPercentDense, loadPly, savePly, increasePcd (both overloads)}
 *                                                      (src/gaussian_model.cpp:48-71, 188-376, 477-510, 553-831, 838-1047, 1090-1100)
 * are extracted VERBATIM, by name, from /root/reference/src/gaussian_model.cpp by oracle/build_ref.py into a generated
 * include file (oracle/_ref/gen/ref_gaussian_model_functions.inc, deleted after the compile) and compiled here against
 * LibTorch.  The only rewrite is the Adam state key: `c10::guts::to_string(param.unsafeGetTensorImpl())` (LibTorch <= 2.1
 * keyed the state by string) -> `param.unsafeGetTensorImpl()` (LibTorch 2.10 keys it by pointer), SURVEY.md 8(b).
 * The reference's include/general_utils.h (inverse_sigmoid, build_rotation) and include/gaussian_parameters.h are included
 * as they are.
 *
 * What is OURS in this file: the class declaration below -- the slice of include/gaussian_model.h:59-193 those functions
 * touch (the full header pulls in Sophus, Eigen, OpenCV, tinyply and ORB-SLAM3 types, none of which exist here), a no-op
 * c10::cuda::CUDACachingAllocator::emptyCache for the host build, and the torch-op glue at the bottom.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref/libref_densify{,_cuda}.so): pins host/src/gaussian_model_densify.cpp, its Python
 * mirror and the HIP stream-compaction kernels behind them to the reference's three-rebuild implementation, Adam-state
 * surgery included.  Two builds of the same source: REF_DENSIFY_DEVICE_CPU maps torch::kCUDA to torch::kCPU (general_utils.h
 * hard-codes kCUDA in build_rotation) so the reference code runs in this GPU-less container; the other build keeps kCUDA
 * (= the HIP device of a ROCm LibTorch) for the GPU boxes.
 */
#include <torch/torch.h>
#include <torch/library.h>

#include <filesystem>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#ifdef REF_DENSIFY_DEVICE_CPU
#define kCUDA kCPU   /* after the torch headers: only the reference's own code below sees it */
#define REF_DEVICE torch::kCPU
#else
#define REF_DEVICE torch::kCUDA
#endif

namespace c10 { namespace cuda { namespace CUDACachingAllocator {
inline void refEmptyCache() {}
}}}
/* src/gaussian_model.cpp:814 calls c10::cuda::CUDACachingAllocator::emptyCache(); the host build has no caching allocator */
#define emptyCache refEmptyCache

#include "general_utils.h"        /* the reference's own headers */
#include "gaussian_parameters.h"
#include "sh_utils.h"             /* RGB2SH for increasePcd */
#include <dlfcn.h>

/* third_party/simple-knn's distCUDA2 (spatial.h:14) for increasePcd (:238,325): OURS -- the reference's is CUDA.  It calls the
 * CPU oracle's gsro_knn (oracle/libgsr_oracle.so, next to this library's directory), which tests/test_reference_pinning.py
 * pins bit for bit to the reference's own simple_knn.cu compiled for the host. */
static torch::Tensor distCUDA2(const torch::Tensor& points)
{
	using knn_fn = void (*)(int, const float*, float*);
	static knn_fn fn = [] {
		Dl_info info;
		TORCH_CHECK(dladdr(reinterpret_cast<void*>(&distCUDA2), &info) && info.dli_fname, "dladdr failed");
		std::string path = std::filesystem::path(info.dli_fname).parent_path().parent_path() / "libgsr_oracle.so";
		void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
		TORCH_CHECK(h, "cannot load ", path, ": ", dlerror());
		auto f = reinterpret_cast<knn_fn>(dlsym(h, "gsro_knn"));
		TORCH_CHECK(f, "gsro_knn not found in ", path);
		return f;
	}();
	auto host = points.detach().to(torch::kCPU).to(torch::kFloat32).contiguous();
	auto out = torch::empty({host.size(0)}, host.options());
	fn(static_cast<int>(host.size(0)), host.data_ptr<float>(), out.data_ptr<float>());
	return out.to(points.device());
}
#define TINYPLY_IMPLEMENTATION    /* third_party/tinyply/tinyply.cpp does exactly this */
#include "third_party/tinyply/tinyply.h"
#include <cstring>
#include <fstream>
#include <iostream>

#define GAUSSIAN_MODEL_TENSORS_TO_VEC                        \
    this->Tensor_vec_xyz_ = {this->xyz_};                    \
    this->Tensor_vec_feature_dc_ = {this->features_dc_};     \
    this->Tensor_vec_feature_rest_ = {this->features_rest_}; \
    this->Tensor_vec_opacity_ = {this->opacity_};            \
    this->Tensor_vec_scaling_ = {this->scaling_};            \
    this->Tensor_vec_rotation_ = {this->rotation_};

/* the members include/gaussian_model.h:59-193 declares and the extracted functions use, same names and types */
class GaussianModel {
public:
	torch::Tensor getScalingActivation();
	torch::Tensor getXYZ();
	torch::Tensor getOpacityActivation();
	void trainingSetup(const GaussianOptimizationParams& training_args);
	void resetOpacity();
	torch::Tensor replaceTensorToOptimizer(torch::Tensor& t, int tensor_idx);
	void prunePoints(torch::Tensor& mask);
	void densificationPostfix(torch::Tensor& new_xyz, torch::Tensor& new_features_dc, torch::Tensor& new_features_rest,
	                          torch::Tensor& new_opacities, torch::Tensor& new_scaling, torch::Tensor& new_rotation,
	                          torch::Tensor& new_exist_since_iter);
	void densifyAndSplit(torch::Tensor& grads, float grad_threshold, float scene_extent, int N = 2);
	void densifyAndClone(torch::Tensor& grads, float grad_threshold, float scene_extent);
	void densifyAndPrune(float max_grad, float min_opacity, float extent, int max_screen_size);
	void addDensificationStats(torch::Tensor& viewspace_point_tensor, torch::Tensor& update_filter);
	float percentDense();
	void setPercentDense(const float percent_dense);
	void loadPly(std::filesystem::path ply_path);
	void savePly(std::filesystem::path result_path);
	void increasePcd(std::vector<float> points, std::vector<float> colors, const int iteration);
	void increasePcd(torch::Tensor& new_point_cloud, torch::Tensor& new_colors, const int iteration);
	torch::Tensor sparse_points_xyz_ = torch::empty({0, 3}), sparse_points_color_ = torch::empty({0, 3});
	int active_sh_degree_ = 0;
	int max_sh_degree_ = 3;

	torch::DeviceType device_type_ = REF_DEVICE;
	torch::Tensor xyz_, features_dc_, features_rest_, scaling_, rotation_, opacity_;
	torch::Tensor max_radii2D_, xyz_gradient_accum_, denom_, exist_since_iter_;
	std::vector<torch::Tensor> Tensor_vec_xyz_, Tensor_vec_feature_dc_, Tensor_vec_feature_rest_, Tensor_vec_opacity_,
	    Tensor_vec_scaling_, Tensor_vec_rotation_;
	std::shared_ptr<torch::optim::Adam> optimizer_;
	float percent_dense_ = 0.01f;
	float spatial_lr_scale_ = 1.0f;
	float lr_init_ = 0.f, lr_final_ = 0.f;
	int lr_delay_steps_ = 0;
	float lr_delay_mult_ = 1.f;
	int max_steps_ = 1000000;
	std::mutex mutex_settings_;
};

#include "ref_gaussian_model_functions.inc"   /* generated by oracle/build_ref.py from src/gaussian_model.cpp, see above */

#undef emptyCache

namespace {

using TensorList = std::vector<torch::Tensor>;

/* params: xyz, features_dc, features_rest, opacity, scaling, rotation (the reference's param-group order);
 * exp_avg / exp_avg_sq: six tensors each (or empty lists: no optimizer state yet); steps: six Adam step counters */
std::unique_ptr<GaussianModel> make_model(const TensorList& params, const TensorList& exp_avg, const TensorList& exp_avg_sq,
                                          const std::vector<int64_t>& steps, torch::Tensor accum, torch::Tensor denom,
                                          torch::Tensor max_radii2D, torch::Tensor exist_since_iter, double percent_dense,
                                          double spatial_lr_scale)
{
	TORCH_CHECK(params.size() == 6, "six parameter tensors expected");
	auto g = std::make_unique<GaussianModel>();
	auto leaf = [](const torch::Tensor& t) { return t.detach().clone().requires_grad_(); };
	g->xyz_ = leaf(params[0]);
	g->features_dc_ = leaf(params[1]);
	g->features_rest_ = leaf(params[2]);
	g->opacity_ = leaf(params[3]);
	g->scaling_ = leaf(params[4]);
	g->rotation_ = leaf(params[5]);
	g->Tensor_vec_xyz_ = {g->xyz_};
	g->Tensor_vec_feature_dc_ = {g->features_dc_};
	g->Tensor_vec_feature_rest_ = {g->features_rest_};
	g->Tensor_vec_opacity_ = {g->opacity_};
	g->Tensor_vec_scaling_ = {g->scaling_};
	g->Tensor_vec_rotation_ = {g->rotation_};
	g->spatial_lr_scale_ = (float)spatial_lr_scale;
	GaussianOptimizationParams opt;
	opt.opacity_lr_ = 0.05f;   /* src/gaussian_parameters.cpp:76 initialises opacity_lr_ with ITSELF (indeterminate); the
	                              mapper overwrites it from its YAML (0.05 in every shipped cfg/gaussian_mapper file) */
	opt.percent_dense_ = (float)percent_dense;
	g->trainingSetup(opt);   /* the reference's: builds torch::optim::Adam with the six groups, zeroes accum / denom */
	g->xyz_gradient_accum_ = accum.detach().clone();
	g->denom_ = denom.detach().clone();
	g->max_radii2D_ = max_radii2D.detach().clone();
	g->exist_since_iter_ = exist_since_iter.detach().clone();
	if (!exp_avg.empty()) {
		TORCH_CHECK(exp_avg.size() == 6 && exp_avg_sq.size() == 6 && steps.size() == 6, "six moment tensors / steps expected");
		auto& state = g->optimizer_->state();
		for (int i = 0; i < 6; i++) {
			auto& param = g->optimizer_->param_groups()[i].params()[0];
			auto s = std::make_unique<torch::optim::AdamParamState>();
			s->step(steps[i]);
			s->exp_avg(exp_avg[i].detach().clone());
			s->exp_avg_sq(exp_avg_sq[i].detach().clone());
			state[param.unsafeGetTensorImpl()] = std::move(s);
		}
	}
	return g;
}

/* -> 6 params, 6 exp_avg, 6 exp_avg_sq (undefined state: zeros-like with step -1), accum, denom, max_radii2D,
 *    exist_since_iter, steps (int64 tensor [6]) */
TensorList dump_model(GaussianModel& g)
{
	TensorList out;
	auto& groups = g.optimizer_->param_groups();
	auto& state = g.optimizer_->state();
	TensorList m, v;
	std::vector<int64_t> steps;
	for (int i = 0; i < 6; i++) {
		auto& param = groups[i].params()[0];
		out.push_back(param.detach());
		auto it = state.find(param.unsafeGetTensorImpl());
		if (it != state.end()) {
			auto& s = static_cast<torch::optim::AdamParamState&>(*it->second);
			m.push_back(s.exp_avg());
			v.push_back(s.exp_avg_sq());
			steps.push_back(s.step());
		} else {
			m.push_back(torch::zeros_like(param));
			v.push_back(torch::zeros_like(param));
			steps.push_back(-1);
		}
	}
	/* the model's own handles must be the optimizer's tensors (GAUSSIAN_MODEL_TENSORS_TO_VEC discipline) */
	TORCH_CHECK(g.xyz_.is_same(groups[0].params()[0]) && g.features_dc_.is_same(groups[1].params()[0]) &&
	            g.features_rest_.is_same(groups[2].params()[0]) && g.opacity_.is_same(groups[3].params()[0]) &&
	            g.scaling_.is_same(groups[4].params()[0]) && g.rotation_.is_same(groups[5].params()[0]),
	            "model tensors and optimizer params diverged");
	out.insert(out.end(), m.begin(), m.end());
	out.insert(out.end(), v.begin(), v.end());
	out.push_back(g.xyz_gradient_accum_);
	out.push_back(g.denom_);
	out.push_back(g.max_radii2D_);
	out.push_back(g.exist_since_iter_);
	out.push_back(torch::tensor(steps, torch::kInt64));
	return out;
}

TensorList ref_densify_and_prune(TensorList params, TensorList exp_avg, TensorList exp_avg_sq, std::vector<int64_t> steps,
                                 torch::Tensor accum, torch::Tensor denom, torch::Tensor max_radii2D,
                                 torch::Tensor exist_since_iter, double percent_dense, double max_grad, double min_opacity,
                                 double extent, int64_t max_screen_size)
{
	torch::NoGradGuard ng;
	auto g = make_model(params, exp_avg, exp_avg_sq, steps, accum, denom, max_radii2D, exist_since_iter, percent_dense, 1.0);
	g->densifyAndPrune((float)max_grad, (float)min_opacity, (float)extent, (int)max_screen_size);
	return dump_model(*g);
}

TensorList ref_reset_opacity(TensorList params, TensorList exp_avg, TensorList exp_avg_sq, std::vector<int64_t> steps,
                             torch::Tensor accum, torch::Tensor denom, torch::Tensor max_radii2D, torch::Tensor exist_since_iter)
{
	torch::NoGradGuard ng;
	auto g = make_model(params, exp_avg, exp_avg_sq, steps, accum, denom, max_radii2D, exist_since_iter, 0.01, 1.0);
	g->resetOpacity();
	return dump_model(*g);
}

TensorList ref_prune_points(TensorList params, TensorList exp_avg, TensorList exp_avg_sq, std::vector<int64_t> steps,
                            torch::Tensor accum, torch::Tensor denom, torch::Tensor max_radii2D, torch::Tensor exist_since_iter,
                            torch::Tensor mask)
{
	torch::NoGradGuard ng;
	auto g = make_model(params, exp_avg, exp_avg_sq, steps, accum, denom, max_radii2D, exist_since_iter, 0.01, 1.0);
	g->prunePoints(mask);
	return dump_model(*g);
}

/* increasePcd (src/gaussian_model.cpp:188-376): vector_overload selects (std::vector<float>, std::vector<float>, int), else
 * (torch::Tensor&, torch::Tensor&, int) */
TensorList ref_increase_pcd(TensorList params, TensorList exp_avg, TensorList exp_avg_sq, std::vector<int64_t> steps,
                            torch::Tensor accum, torch::Tensor denom, torch::Tensor max_radii2D, torch::Tensor exist_since_iter,
                            torch::Tensor points, torch::Tensor colors, int64_t iteration, bool vector_overload)
{
	torch::NoGradGuard ng;
	auto g = make_model(params, exp_avg, exp_avg_sq, steps, accum, denom, max_radii2D, exist_since_iter, 0.01, 1.0);
	if (vector_overload) {
		auto p = points.detach().to(torch::kCPU).contiguous(), c = colors.detach().to(torch::kCPU).contiguous();
		std::vector<float> pv(p.data_ptr<float>(), p.data_ptr<float>() + p.numel()), cv(c.data_ptr<float>(), c.data_ptr<float>() + c.numel());
		g->increasePcd(pv, cv, (int)iteration);
	} else {
		auto p = points.detach().clone(), c = colors.detach().clone();
		g->increasePcd(p, c, (int)iteration);
	}
	return dump_model(*g);
}

/* addDensificationStats (src/gaussian_model.cpp:817-831) on a viewspace tensor whose .grad() is `viewspace_grad` */
TensorList ref_add_densification_stats(torch::Tensor accum, torch::Tensor denom, torch::Tensor viewspace_grad,
                                       torch::Tensor update_filter)
{
	GaussianModel g;
	g.xyz_gradient_accum_ = accum.detach().clone();
	g.denom_ = denom.detach().clone();
	auto vs = torch::zeros_like(viewspace_grad).requires_grad_();
	vs.mutable_grad() = viewspace_grad.detach().clone();
	{
		torch::NoGradGuard ng;
		g.addDensificationStats(vs, update_filter);
	}
	return {g.xyz_gradient_accum_, g.denom_};
}

/* One optimizer step of the reference's own setup: trainingSetup (:477-510) + the given learning rate for xyz (what
 * updateLearningRate installs) + torch::optim::Adam::step on the given gradients.  Pins gsr_adam_step and the
 * learning-rate / eps / group layout of GaussianModel::trainingSetup against the C++ optimizer the reference runs. */
TensorList ref_adam_step(TensorList params, TensorList grads, TensorList exp_avg, TensorList exp_avg_sq,
                         std::vector<int64_t> steps, double spatial_lr_scale, double xyz_lr)
{
	auto z = torch::zeros({params[0].size(0), 1}, params[0].options());
	auto g = make_model(params, exp_avg, exp_avg_sq, steps, z, z, z.squeeze(1), z.squeeze(1).to(torch::kInt32), 0.01,
	                    spatial_lr_scale);
	if (xyz_lr >= 0) g->optimizer_->param_groups()[0].options().set_lr(xyz_lr);
	for (int i = 0; i < 6; i++)
		if (grads[i].defined() && grads[i].numel()) g->optimizer_->param_groups()[i].params()[0].mutable_grad() = grads[i].detach().clone();
	g->optimizer_->step();
	return dump_model(*g);
}

}  // namespace

/* GaussianModel::savePly / loadPly of the reference (tinyply): the checkpoint format is pinned against THEIR writer / reader */
void ref_save_ply(TensorList params, std::string path)
{
	auto z1 = torch::zeros({params[0].size(0), 1}, params[0].options());
	auto g = make_model(params, {}, {}, {}, z1, z1, z1.squeeze(1), z1.squeeze(1).to(torch::kInt32), 0.01, 1.0);
	g->savePly(path);
}
TensorList ref_load_ply(std::string path, int64_t max_sh_degree)
{
	GaussianModel g;
	g.max_sh_degree_ = (int)max_sh_degree;
	g.loadPly(path);
	return {g.xyz_, g.features_dc_, g.features_rest_, g.opacity_, g.scaling_, g.rotation_, torch::tensor({(int64_t)g.active_sh_degree_})};
}

#define REF_MODEL_OPS(m)                                              \
	m.def("save_ply", &ref_save_ply);                                 \
	m.def("load_ply", &ref_load_ply);                                 \
	m.def("densify_and_prune", &ref_densify_and_prune);               \
	m.def("increase_pcd", &ref_increase_pcd);                         \
	m.def("reset_opacity", &ref_reset_opacity);                       \
	m.def("prune_points", &ref_prune_points);                         \
	m.def("add_densification_stats", &ref_add_densification_stats);   \
	m.def("adam_step", &ref_adam_step);

#ifdef REF_DENSIFY_DEVICE_CPU
TORCH_LIBRARY(photoslam_reference_model, m) { REF_MODEL_OPS(m) }
#else
TORCH_LIBRARY(photoslam_reference_model_cuda, m) { REF_MODEL_OPS(m) }
#endif
