// gaussian_model_lite.h -- the slice of GaussianModel / GaussianKeyframe the measured train step
// needs (include/gaussian_model.h:59-193, include/gaussian_keyframe.h:36-136 of the reference),
// LibTorch only -- the reference classes pull in OpenCV, Eigen, Sophus and ORB-SLAM3, none of which
// exist in this environment.  Member names follow the reference so GaussianRenderer::render is the
// same code for both.
#pragma once
#include <torch/torch.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include "rasterize_points.h"   // RasterWorkspace

#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "rasterize_points.h"   // ShAdamStep

struct GaussianOptimizationParams {  // include/gaussian_parameters.h:61-96 defaults
	int iterations_ = 30000;
	float position_lr_init_ = 0.00016f, position_lr_final_ = 0.0000016f, position_lr_delay_mult_ = 0.01f;
	int position_lr_max_steps_ = 30000;
	float feature_lr_ = 0.0025f, opacity_lr_ = 0.05f, scaling_lr_ = 0.005f, rotation_lr_ = 0.001f;
	float percent_dense_ = 0.01f, lambda_dssim_ = 0.2f;
	int densification_interval_ = 100, opacity_reset_interval_ = 3000, densify_from_iter_ = 500,
	    densify_until_iter_ = 15000;
	float densify_grad_threshold_ = 0.0002f;
};

#ifndef GSR_HAVE_PIPELINE_PARAMS
#define GSR_HAVE_PIPELINE_PARAMS
struct GaussianPipelineParams {   // include/gaussian_parameters.h:41-49
	bool convert_SHs_ = false;
	bool compute_cov3D_ = false;
};
#endif

struct GaussianKeyframe {
	int image_height_ = 0, image_width_ = 0;
	float FoVx_ = 0.f, FoVy_ = 0.f;
	torch::Tensor world_view_transform_, full_proj_transform_, camera_center_;
};

// One Adam parameter group of the fused optimizer (gsr_adam_step)
struct AdamGroup {
	torch::Tensor param, exp_avg, exp_avg_sq;
	double lr = 0.0, lr_tail = 0.0;   // torch::optim::AdamOptions::lr is double (set_lr(float) widens: src/gaussian_model.cpp:489-502)
	int period = 0, split = 0;
	int step = 0;   // per parameter, as torch::optim::AdamParamState: advances only when the parameter has a gradient
};

class GaussianModel {
public:
	GaussianModel(int sh_degree, torch::Tensor xyz, torch::Tensor features, torch::Tensor opacity, torch::Tensor scaling,
	              torch::Tensor rotation, float spatial_lr_scale);
	explicit GaussianModel(int sh_degree);   // empty model: createFromPcd() next

	// map maintenance (src/gaussian_model_densify.cpp; include/gaussian_model.h:72-137 of the reference)
	void createFromPcd(torch::Tensor points, torch::Tensor colors, float spatial_lr_scale);
	// src/gaussian_model.cpp:188-376, both overloads: new SLAM map points join the model (colours -> SH DC term, scales from
	// distCUDA2 AMONG THE NEW POINTS, identity rotations, opacity 0.1, exist_since_iter = iteration) through
	// densificationPostfix (:644-712): appended behind the existing rows, zero Adam moments for the new rows, step counters
	// carried, all three statistics arrays reset to zero.  The append is O(new points) while the arena has room (the reference
	// re-cats every tensor and moment and then empties the allocator cache).
	void increasePcd(std::vector<float> points, std::vector<float> colors, const int iteration);
	void increasePcd(torch::Tensor& new_point_cloud, torch::Tensor& new_colors, const int iteration);
	void oneUpShDegree();
	// Loop closure (src/gaussian_model.cpp:379-475; callers: src/gaussian_mapper.cpp loop-closure handling).
	// applyScaledTransformation: every point p <- s R p + t (transformPoints on the scaled positions), then -- exactly as
	// shipped -- `scaling_ *= s` on the LOG-scales (:395; not log(s) added: mirrored, not fixed), and the Adam moments of xyz and
	// scaling are zeroed, their step counters kept (scaledTransformationPostfix -> replaceTensorToOptimizer :567-586).
	// T: the 4x4 matrix [R t; 0 1], row-major (the reference takes a Sophus::SE3f and hands its matrix on).
	void applyScaledTransformation(const float s, torch::Tensor T);
	// The points a keyframe sees and that exist for fewer than stable_num_iter_existence iterations around its creation are
	// moved by diff_pose (scaleAndTransformThenMarkVisiblePoints: positions AND rotations), flags of moved points are cleared,
	// rotation_ becomes the NORMALISED (and rotated) quaternions, moments of xyz and rotation are zeroed (:408-475).
	void scaledTransformVisiblePointsOfKeyframe(torch::Tensor& point_not_transformed_flags, torch::Tensor& diff_pose,
	                                            torch::Tensor& kf_world_view_transform, torch::Tensor& kf_full_proj_transform,
	                                            const int kf_creation_iter, const int stable_num_iter_existence, int& num_transformed,
	                                            const float scale = 1.0f);
	void resetOpacity();
	bool intended_opacity_reset_ = false;   // see resetOpacity(): false = the reference as shipped
	void prunePoints(torch::Tensor& mask);
	// the whole model laid out along a Z-order curve of the positions (gaussian_model_densify.cpp); returns the permutation
	torch::Tensor reorderAlongZCurve();
	struct DensifyResult {
		int64_t cloned = 0, split = 0, pruned = 0, points = 0;
	};
	// generator: the sampler of the split children's positions (every data-parallel rank seeds it identically)
	DensifyResult densifyAndPrune(float max_grad, float min_opacity, float extent, int max_screen_size,
	                              c10::optional<at::Generator> generator = c10::nullopt);

	// activations, src/gaussian_model.cpp:48-71
	torch::Tensor getXYZ() { return xyz_; }
	torch::Tensor getScalingActivation() { return torch::exp(scaling_); }
	torch::Tensor getRotationActivation() { return torch::nn::functional::normalize(rotation_); }
	torch::Tensor getOpacityActivation() { return torch::sigmoid(opacity_); }
	// one [P,16,3] leaf instead of cat(features_dc.clone(), features_rest.clone()) (gaussian_model.cpp:63-66)
	// (lazy SH Adam: outside the fused train step the rows are brought up to date first, see syncFeatures())
	torch::Tensor getFeatures()
	{
		if (!in_lazy_step_) syncFeatures();
		return features_;
	}
	torch::Tensor getCovarianceActivation(int scaling_modifier = 1);   // :73-96

	void trainingSetup(const GaussianOptimizationParams& opt);   // src/gaussian_model.cpp:477-510
	float updateLearningRate(int step);                          // :1118-1131 (exponLrFunc)
	void optimizerStep();                                        // torch::optim::Adam semantics, fused
	// the same step one parameter group at a time (xyz, features, opacity, scaling, rotation), so that a data-parallel
	// driver can update a tensor as soon as ITS gradient reduction has landed
	void beginOptimizerStep() {}   // (the step counters are per group and advance in optimizerStepGroup)
	void optimizerStepGroup(int group);
	void zeroGrad();
	void addDensificationStats(torch::Tensor& viewspace_point_tensor, torch::Tensor& update_filter);  // :817-831
	std::vector<torch::Tensor> params()
	{
		syncFeatures();
		return {xyz_, features_, opacity_, scaling_, rotation_};
	}
	// the five leaves as they are (no catch-up of lazily stepped SH rows): for code that only needs their .grad() inside a step
	std::vector<torch::Tensor> paramsRaw() { return {xyz_, features_, opacity_, scaling_, rotation_}; }

	// Lazy Adam steps for the SH rows of culled Gaussians (gsr_sh_adam_lazy, include/gsr.h; TrainStep::lazy_sh_adam_window_).
	// While features_row_step_ is defined, rows of features_ and of its two moment tensors may be up to `window` zero-gradient
	// steps behind (row i has taken features_row_step_[i] of groups_[1].step steps).  syncFeatures() takes the missing steps
	// (gsr_sh_adam_flush: the same arithmetic, bit-identical to the eager update) and drops the state; everything that reads
	// or rewrites features_ or its moments outside the fused train step calls it first (getFeatures(), params(), the dense
	// optimizer step, densify / prune, savePly, the data-parallel exchange).
	void syncFeatures();
	torch::Tensor features_row_step_;                            // [P] int32 or undefined (= every row is up to date)
	std::vector<std::pair<double, double>> features_lr_hist_;    // (lr, lr_tail) of the Adam steps since the state exists, newest first
	int features_lazy_window_ = 0;
	bool in_lazy_step_ = false;                                  // set by TrainStep around its own render call

	int max_sh_degree_, active_sh_degree_;
	float spatial_lr_scale_;
	// Multiplies every learning rate where it is consumed.  1 = training.  0 = a STATIONARY workload for throughput
	// measurements: every kernel runs and the Adam moments update, the parameters do not move (bench.py).
	double lr_scale_ = 1.0;
	torch::Tensor xyz_, features_, opacity_, scaling_, rotation_;
	torch::Tensor max_radii2D_, xyz_gradient_accum_, denom_;
	// include/gaussian_model.h:169: the iteration each Gaussian has existed since ([P] int32; read by the loop-closure transform,
	// src/gaussian_model.cpp:433); carried through prune / clone / split (:636, :744, :782) and set by increasePcd (:255)
	torch::Tensor exist_since_iter_;
	torch::Tensor sparse_points_xyz_, sparse_points_color_;   // :182-183: the points handed to increasePcd so far
	GaussianOptimizationParams opt_;
	std::vector<AdamGroup> groups_;

	// arena of the rebuilds (gaussian_model_densify.cpp): optional, created on the first rebuild otherwise
	void reserve(int64_t capacity);
	// checkpoint interchange with the reference and with Inria viewers (src/ply_io.cpp; src/gaussian_model.cpp:838-1047)
	void savePly(const std::string& result_path);
	void loadPly(const std::string& ply_path);
	torch::Device device_ = torch::kCPU;   // where loadPly() / createFromPcd() put a model that has no tensors yet

	// densifyAndPrune lays the new set out along a Z-order curve of the Gaussians' positions (include/gsr.h:
	// gsr_densify_gather_args.morton_scratch): the same Gaussians, values, moments and statistics in another row order -- the
	// per-Gaussian kernels of the following steps then find a view's Gaussians in shared 128-byte lines.  Off by default (the
	// reference's order: kept originals, clones, children).
	bool morton_reindex_ = false;

private:
	torch::Tensor& paramByIndex(int i);
	void replaceParam(int group, torch::Tensor fresh, torch::Tensor exp_avg, torch::Tensor exp_avg_sq, bool rows_kept = false);
	void preloadMaintenanceKernels();   // first-use code-object loads of the ATen kernels behind resetOpacity / loop closure, paid in trainingSetup
	void replaceParamValues(int group, torch::Tensor fresh);   // same shape, zero moments: in place while the leaf lives in the arena
	// gsr_densify_select + one host read + gsr_densify_gather; returns kept, clones, child parents, split, clone-selected, rows
	std::array<int64_t, 6> compact(struct gsr_densify_select_args& sel, c10::optional<at::Generator> generator, bool morton_reindex = false);
	static void* hostStream(const torch::Tensor& t);   // the current HIP stream of the tensor's device (null on the host)
	struct Arena {
		int64_t capacity = 0;
		int cur = 0;
		std::array<std::array<torch::Tensor, 3>, 5> params[2];   // [set][tensor][parameter, exp_avg, exp_avg_sq]
		std::array<torch::Tensor, 3> stats[2];
		torch::Tensor exist[2];   // exist_since_iter_ ([capacity] int32)
	} arena_;
	void appendRows(const std::array<torch::Tensor, 5>& rows, int iteration);   // densificationPostfix as an append into the arena
public:
	// The arena keeps two sets of 1.25-1.5x (parameters + moments) resident (~2 kB per Gaussian) so that rebuilds allocate
	// nothing.  LIFETIME CONTRACT: after a rebuild xyz_, features_, ..., the Adam moments and the statistics are narrow() views
	// into one of the two sets; a tensor obtained BEFORE rebuild N (getXYZ(), a moment, xyz_gradient_accum_ held by a viewer or
	// mapper thread) is overwritten by rebuild N + 2 -- clone() what has to outlive a rebuild (the reference hands out fresh
	// tensors).  releaseArena(): the live tensors move into allocations of their own and both sets are freed (what the
	// reference's emptyCache() after densification achieves, :814) -- call it when densification ends
	// (iteration >= densify_until_iter_) or whenever stable tensors are needed; the next rebuild creates a new arena.
	void releaseArena();
private:
	torch::Tensor densify_scratch_;
};

// GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:614-774) without the SLAM keyframe
// scheduling: render -> masked L1 + lambda (1 - SSIM) -> backward -> statistics -> Adam.
class TrainStep {
public:
	TrainStep(std::shared_ptr<GaussianModel> g, torch::Tensor background) : gaussians_(g), background_(background) {}
	~TrainStep();                                  // releases the HIP events of the data-parallel step (keyframe_batch_exchange.cpp)
	TrainStep(const TrainStep&) = delete;          // (it owns them)
	TrainStep& operator=(const TrainStep&) = delete;
	// forward + backward only (gradients left on the leaves, statistics gathered): lets a data-parallel
	// driver all-reduce before finishOneIteration()
	torch::Tensor renderAndBackward(std::shared_ptr<GaussianKeyframe> kf, torch::Tensor gt_image, torch::Tensor mask);
	void finishOneIteration();
	// finishOneIteration() in three pieces for the overlapped data-parallel step (bench.py): statistics, then Adam per
	// group as the reductions complete, then the gradient reset
	void finishBegin();
	void finishAdamGroup(int group);
	void finishEnd();
	// Keyframe batches over several ranks: every rank accumulates the statistics of ITS views (they are added inside
	// backward); SUM / MAX commute with that accumulation, so the driver reduces xyz_gradient_accum_ and denom_ (SUM) and
	// max_radii2D_ (MAX) over the ranks only before a finishBegin() that will densify -- densifyDue() says when.
	bool densifyDue() const;
	torch::Tensor trainForOneIteration(std::shared_ptr<GaussianKeyframe> kf, torch::Tensor gt_image, torch::Tensor mask)
	{
		if (process_group_) return trainForOneIterationDataParallel(kf, gt_image, mask);
		auto loss = renderAndBackward(kf, gt_image, mask);
		finishOneIteration();
		return loss;
	}
	// Keyframe batches, one keyframe per rank: with a process group set, trainForOneIteration() is the
	// data-parallel step -- render + backward of THIS rank's keyframe, the gradient exchange over c10d (RCCL on the GPU boxes:
	// ViewFactoredExchange by default, the plain GradientReduction otherwise; host/include/keyframe_batch_exchange.h), the
	// optimizer on the batch-mean gradient, and before a densification the SUM / MAX of the per-rank statistics.  Every rank
	// holds a replica; replicas stay bit-identical (identical Adam on identical gradients, identically seeded split samples).
	// No collective is issued from anywhere but here.  A C++ mapper creates c10d::TCPStore + ProcessGroupNCCL and hands the
	// group over; under Python the default group is resolved by name (ops_register.cpp: trainer_set_process_group).
	void setProcessGroup(c10::intrusive_ptr<c10d::ProcessGroup> pg, bool factored = true);
	torch::Tensor trainForOneIterationDataParallel(std::shared_ptr<GaussianKeyframe> kf, torch::Tensor gt_image, torch::Tensor mask);
	c10::intrusive_ptr<c10d::ProcessGroup> process_group_;
	// The all-gather of the view-factored exchange is issued on a stream of its own that waits only for the point inside
	// backward at which the colour gradients are complete (gsr_backward_args.color_view_ready_stream): it overlaps the SH
	// backward kernel instead of following it.  A hipStream_t from LibTorch's pool, created on first use; null on the host.
	void* gather_stream_ = nullptr;
	bool gather_stream_in_use_ = false;   // this step hands gather_stream_ to the exchange (early_gather_)
	// Data-parallel keyframe batches with the view-factored exchange (include/gsr.h, gsr_sh_grad_from_views): backward
	// then leaves the clamp-masked colour gradient of this view in sh_grad_view_ and no gradient on features_; after the
	// driver has gathered the views of all ranks, setFeaturesGradFromViews() installs the batch-mean SH gradient.  It reads
	// xyz_, so it must run before finishAdamGroup(0).
	// Densification schedule of GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:711-735), off by default:
	// every densification_interval_ iterations after densify_from_iter_, and the opacity reset.  Runs inside
	// finishBegin(); an iteration that rebuilt the tensors skips its optimizer step exactly as the reference does (the
	// fresh leaves have no gradient).
	bool densify_ = false;
	// The step of the position learning-rate schedule for the NEXT iteration(s): -1 = the iteration count (the COLMAP flavour,
	// src/gaussian_mapper.cpp:672-674); a SLAM session sets the keyframe's use count before every iteration (:663-671:
	// kfs_used_times_[fid], capped at position_lr_max_steps_)
	int position_lr_step_ = -1;
	float cameras_extent_ = 1.0f, densify_min_opacity_ = 0.005f;
	int prune_big_point_after_iter_ = 30000;   // Optimization.prune_big_point_after_iter of the shipped Replica / EuRoC configs (no in-code default in the reference)
	c10::optional<at::Generator> generator_;
	GaussianModel::DensifyResult last_densify_;
	// The Adam step of the SH tensor inside the rasterizer's backward (gsr_backward_args.sh_adam: its gradient rows never
	// reach HBM, 0.13 ms of a 2.3 ms step at C3); same arithmetic as the separate pass.  Not used on an iteration that
	// densifies (the reference skips that optimizer step) nor with the factored exchange.
	bool fused_sh_adam_ = true;
	// ... and the zero-gradient steps of the culled Gaussians' rows taken lazily, at most this many at a time (2 .. 32; 0 = every
	// row at every step): one HBM round trip of a culled row per `window` steps instead of one per step (1.2 GB per step at
	// C3), results bit-identical (GaussianModel::syncFeatures, tests/test_lazy_sh_adam.py)
	int lazy_sh_adam_window_ = 32;
	// The Adam steps of xyz / opacity / scaling / rotation inside the backward kernels that hold their gradients
	// (gsr_backward_args.geom_adam): the 88 B per Gaussian gradient round trip and four optimizer launches disappear; on the
	// same iterations as fused_sh_adam_.  The viewspace gradient and dL_dcov3D are then not written either.
	bool fused_geom_adam_ = true;
	bool factored_exchange_ = false;
	// Scheduling / list-building switches, each at its measured-best value; none changes a result.  They
	// are options of THIS object -- a SLAM process that links the library decides per TrainStep -- and the environment variables
	// of the bench sessions (GSR_CULL_EMPTY_TILES, GSR_EARLY_GATHER, GSR_LAZY_SLICE_EARLY, GSR_SH_ADAM_SIDE_STREAM) only override.
	bool cull_empty_tiles_ = false;   // instances of tiles no pixel of which can blend the Gaussian leave the list (a wash on MI355X)
	// the rasterizer's three scratch buffers, kept across iterations and grown with headroom (rasterize_points.h: RasterWorkspace);
	// persistent_workspace_ = false: fresh buffers per call, as the reference's resizeFunctional
	bool persistent_workspace_ = true;
	RasterWorkspace workspace_;
	bool early_gather_ = true;        // the exchange's all-gather waits for the colour gradients only, not for the whole backward pass
	// The view-factored exchange in its PACKED form (include/gsr.h: gsr_pack_color_view): every rank sends only the rows its
	// view sees -- 11.7 MB instead of 24 MB per rank and link at 2 M Gaussians.  The ranks agree on the message capacity by
	// exchanging their views' visible counts (one int each, host values) over a host-side group (setCountGroup: gloo) right behind
	// the forward pass; the host picks the result up after it has queued the backward pass (keyframe_batch_exchange.cpp:
	// beginCountExchange).  Off by default until a multi-GPU node has measured it (bench.py picks the faster form in a guarded
	// trial); bit-identical results (tests/test_packed_views.py, tests/test_train_step.py).
	bool packed_exchange_ = false;
	bool lazy_slice_late_ = false;    // the lazy SH rows' slice behind the backward blend instead of next to it
	bool no_side_stream_ = false;     // no second stream inside gsr_forward / gsr_backward
	// exposed communication of the data-parallel step: the time the compute stream spent waiting for a collective (HIP events
	// around every Work::wait(), recorded only while profile_exchange_ is set); read with exchangeWaitMs()
	bool profile_exchange_ = false;
	std::vector<double> exchangeWaitMs();
	void markWait(int k);   // {all-gather wait, all-reduce wait} of the last data-parallel step, in ms
	// pipeline flags of the render call (include/gaussian_parameters.h; every shipped config leaves both off).  The fused
	// optimizer paths above are taken only when render() keeps the tensors they step inside the rasterizer.
	GaussianPipelineParams pipe_;
	torch::Tensor sh_send_;        // [P + 1, 3]: rows 0 .. P-1 = sh_grad_view_, row P = this view's camera centre (one all-gather)
	// [N, P + 1, 3]: what the all-gather writes.  PERSISTENT (re-allocated only when P or N changes) and sized in
	// renderAndBackward() BEFORE the forward and backward passes are enqueued: the early gather writes it from a second stream
	// that is ordered only behind "the colour gradients are complete", i.e. while the tail of the backward pass still runs -- a
	// buffer taken from the caching allocator at that point could be a block the pass has just released and still writes
	// (record_stream protects the free side, not the first use on a foreign stream).
	torch::Tensor sh_gathered_;
	// packed form: the message buffers (int32; persistent like sh_gathered_), the pack kernels' scratch and the count exchange
	torch::Tensor sh_packed_send_, sh_packed_gathered_, sh_pack_scratch_;
	// counts_host_: every rank's visible count, on the host (pinned only on the RCCL route, whose copies need it)
	torch::Tensor count_own_pinned_, count_own_dev_, counts_dev_, counts_host_, count_own_host_;
	// the visible counts travel host-side over this group when it is set (gloo next to the RCCL group of the gradients):
	// keyframe_batch_exchange.cpp: beginCountExchange
	c10::intrusive_ptr<c10d::ProcessGroup> count_group_;
	void setCountGroup(c10::intrusive_ptr<c10d::ProcessGroup> pg) { count_group_ = std::move(pg); }
	c10::intrusive_ptr<c10d::Work> count_work_;
	bool counts_on_device_route_ = false;
	void* counts_event_ = nullptr;     // hipEvent_t behind the counts' copy to the host
	bool packed_this_step_ = false;
	// the gathered messages' headers are read back and verified (checkPackedViews: a host wait) on the first packed steps of this
	// object -- a capacity the ranks did not agree on shows there, loudly, instead of as rows silently read as zero
	int check_packed_first_steps_ = 3;
	int packed_steps_checked_ = 0;
	// On = the backward pass writes the message itself (gsr_backward_args.packed_view; mask + prefix planned from the radii on the
	// gather stream behind the forward pass): no pack launches between the backward pass and the gather.  Off (default) = the
	// message is packed from the dense view behind the backward pass (gsr_pack_color_view, four launches on the gather stream next
	// to the SH backward kernel).  Measured on one box at one rank (profiles/r04_r): writing the message inside the backward pass
	// costs the compute stream +26 us, packing behind it costs +3 us there (it hides next to the SH kernel) -- the same step time
	// either way, so the simpler arrangement is the default.  GSR_PACK_IN_BACKWARD overrides.
	bool pack_in_backward_ = false;
	bool prepacked_this_step_ = false;
	void planPackedView(const torch::Tensor& radii);
	void beginCountExchange();         // behind the forward pass: this view's visible count to every rank
	int64_t finishCountExchange();     // -> the capacity every rank uses (max count, rounded up to 4 rows)
	void stepFeaturesFromPackedViews(torch::Tensor messages, int64_t msg_stride, int64_t n_views);
	// hipEvent_t: before / after the gather wait, before / after the reduce wait, start and end of the data-parallel step
	void* wait_events_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	torch::Tensor sh_grad_view_;
	void setFeaturesGradFromViews(torch::Tensor campos_views, torch::Tensor dL_dcolor_views);
	// ... or rebuilds it and takes the Adam step of features_ in the same pass (gsr_sh_adam_from_views): the mean gradient
	// never reaches HBM.  Replaces setFeaturesGradFromViews() + finishAdamGroup(1); same ordering constraint.
	// row0 / first_part: the gathered views may arrive in parts (rows [row0, row0 + views.size(1)) of the Gaussians): each
	// part is applied as soon as ITS all-gather has landed, while the next one is still on the links; the Adam step counter
	// advances with the first part only.
	void stepFeaturesFromViews(torch::Tensor campos_views, torch::Tensor dL_dcolor_views, int64_t row0 = 0, bool first_part = true);
	// With lazy_sh_adam_window_ >= 2 the factored step is as lean as the single-GPU one: a row no gathered view lights takes a
	// zero-gradient step, i.e. it may take it LATER (gsr_sh_adam_from_views with sh_adam->lazy) -- renderAndBackward() then
	// advances the SH step counter itself and hands the lazy struct to the forward pass, stepFeaturesFromViews() steps only the
	// lit rows, and finishFeaturesFromViews() -- after the last part -- runs this step's slice of the rotating catch-up and
	// records the step's learning rates.  (finishEnd() calls it if the driver did not.)
	void finishFeaturesFromViews();
	// xyz / opacity / scaling / rotation in ONE Adam launch (gsr_adam_step_multi): the data-parallel step's four small
	// gradients arrive together from one all-reduce.  Replaces finishAdamGroup(0 / 2 / 3 / 4).  grad_scale: the gradients are
	// multiplied by it as they are read (the 1/N of a batch mean whose all-reduce summed).
	void finishGeomAdam(float grad_scale = 1.0f);
	ShAdamStep views_adam_;
	bool views_adam_pending_ = false;
	std::shared_ptr<GaussianModel> gaussians_;
	torch::Tensor background_;
	int iteration_ = 0;
	torch::Tensor last_viewspace_, last_visibility_ /* undefined when the step fused its consumers: last_radii_ > 0 */, last_radii_;
	torch::Tensor root_grad_;   // the constant 1 handed to loss.backward()
	struct MaskEntry {
		c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl> self{c10::intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>()};
		int64_t version = 0;
		bool ones = false;
	};
	std::map<const void*, MaskEntry> mask_is_ones_;   // masks seen so far: all ones? (renderAndBackward)
};

// loss = (1-lambda) L1 + lambda (1-SSIM) with its gradient in two HIP kernels (gsr_l1_ssim_loss)
// is_root: the caller promises to call backward() on this very value (upstream gradient exactly 1, as TrainStep does):
// backward then hands the stored gradient on without the [3,H,W] multiply by one
torch::Tensor fusedL1SSIMLoss(torch::Tensor rendered, torch::Tensor gt, torch::Tensor mask, float lambda_dssim,
                              bool is_root = false);
