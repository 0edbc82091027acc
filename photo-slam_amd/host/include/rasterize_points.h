// rasterize_points.h -- LibTorch boundary of the MI355X rasterizer.  RasterizeGaussiansCUDA, RasterizeGaussiansBackwardCUDA and
// markVisible are declared with EXACTLY the parameter lists of the reference's include/rasterize_points.h:18-65: the
// same mangled symbols, so an object compiled against the reference header links against libphotoslam_host.so
// (tests/test_reference_link.py does that).  The extensions of this repository are separate OVERLOADS with extra
// parameters (only the last one, the workspace of the longest overload, has a default).  On a ROCm build of LibTorch torch::kCUDA *is* the HIP device.
// Implementation: src/rasterize_points.cpp on top of the C-ABI in include/gsr.h (libgsr_hip.so).
#pragma once
#include <torch/torch.h>

#include <tuple>
#include <vector>

// Extension, optimizer-in-backward for the SH tensor (gsr_sh_adam of include/gsr.h): when exp_avg is defined, backward applies
// this Adam step to `sh` IN PLACE instead of computing dL_dsh (which then comes back undefined).
// Lazy mode (gsr_sh_adam_lazy): with row_step defined the rows of culled Gaussians take their zero-gradient steps later,
// several at a time -- the SAME struct then goes to the forward overload below (rows that become visible are brought up to
// date before they are evaluated) and to backward, and shAdamFlush must run before anything else touches sh or the moments.
struct ShAdamStep {
	torch::Tensor exp_avg, exp_avg_sq;   // [P,16,3], contiguous
	double lr = 0.0, lr_tail = 0.0, beta1 = 0.9, beta2 = 0.999, eps = 1e-15;   // double, as torch::optim::AdamOptions
	int step = 0;
	torch::Tensor row_step;              // lazy mode: [P] int32, the Adam steps each row has taken; undefined = eager
	int window = 0;                      // lazy mode: 2 .. GSR_SH_LAZY_WINDOW
	std::vector<double> lr_past, lr_tail_past;   // lazy mode: [k-1] = the learning rates of step (step - k)
	// view-factored mode only (gsr_backward_args.color_view_ready_stream; consulted by backward, with or without the fields
	// above): a hipStream_t that is made to wait for the point inside backward at which dL_dcolor_view is complete -- the
	// exchange issues its all-gather there and overlaps the last kernel of the pass
	void* color_view_ready_stream = nullptr;
	// view-factored mode, packed exchange (gsr_backward_args.packed_view): an int32 message whose mask / prefix sections
	// packViewPlan() fills between the forward and the backward pass; backward writes rows and header next to dL_dcolor_view
	torch::Tensor packed_view;
	int64_t packed_capacity = 0;
	// scheduling of the optimizer work the library forks next to its own kernels (gsr_sh_adam: zero = the measured-best
	// arrangement; the environment variables GSR_SH_ADAM_SIDE_STREAM / GSR_LAZY_SLICE_EARLY / GSR_SH_ADAM_SIDE_BLOCKS override)
	bool no_side_stream = false, lazy_slice_late = false;
	int side_blocks = 0;
};

// (num_rendered, out_color[3,H,W], radii[P] i32, geomBuffer u8, binningBuffer u8, imgBuffer u8)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered);
// overload with the extension parameter raw_params: GSR_RAW_* mask, see include/gsr.h
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered, const int raw_params);
// Persistent scratch for a caller that renders iteration after iteration (TrainStep): the three byte buffers of the rasterizer
// (the reference allocates them per call, src/rasterize_points.cu:71-76, and returns them).  They grow with 50 % headroom and
// never shrink.  The binning buffer's size follows the instance count, which changes with every step of a training run: per-call
// allocations of ever-new sizes leave the caching allocator with a trail of blocks none of which fits the next request
// (measured over 300 mapper iterations at 4 M Gaussians: 46 GB reserved for 10 GB in use against 15 GB with the workspace, and
// every new largest size is a hipMalloc of a gigabyte -- up to 45 ms on the pool's boxes, profiles/r06_w*).
struct RasterWorkspace {
	torch::Tensor geom, binning, img;
};

// ... and the lazy SH Adam state (only consulted when sh_adam.row_step is defined; `sh` is then updated in place); workspace:
// nullptr = fresh buffers per call, as the reference
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> RasterizeGaussiansCUDA(
    const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
    const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
    const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
    const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
    const int image_width, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
    const bool prefiltered, const int raw_params, const ShAdamStep& sh_adam, RasterWorkspace* workspace = nullptr);

// Extension, optimizer-in-backward for xyz / opacity / scaling / rotation (gsr_geom_adam of include/gsr.h): when param is
// filled (four entries each, in that order), backward applies this Adam step to the four tensors IN PLACE instead of computing
// their gradients (which then come back undefined).  Needs raw_params == 7 and scales / rotations (no cov3D_precomp).
// training_outputs_only: dL_dmeans2D and dL_dcov3D are not written either (undefined) -- for a caller that fuses the
// densification statistics (view_stats).
struct GeomAdamStep {
	std::vector<torch::Tensor> param, exp_avg, exp_avg_sq;   // xyz [P,3], opacity [P,1], scaling [P,3], rotation [P,4]
	std::vector<double> lr;
	std::vector<int64_t> step;
	double beta1 = 0.9, beta2 = 0.999, eps = 1e-15;
	bool training_outputs_only = false;
};

// (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer);
// overload with the extension parameters (all four, no defaults)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                               const int raw_params,
                               /* a [P,3] float tensor that receives the clamp-masked colour gradient; dL_dsh is then NOT
                                  computed and comes back undefined (gsr_backward_args.dL_dcolor_view); undefined = off */
                               const torch::Tensor& dL_dcolor_view,
                               const ShAdamStep& sh_adam,
                               /* {xyz_gradient_accum, denom, max_radii2D} (P floats each), updated in place with this
                                  view's densification statistics (gsr_backward_args.stat_*); empty = off */
                               const std::vector<torch::Tensor>& view_stats);
// ... and the fused geometry step (GeomAdamStep above)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp,
                               const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const float tan_fovx,
                               const float tan_fovy, const torch::Tensor& dL_dout_color, const torch::Tensor& sh,
                               const int degree, const torch::Tensor& campos, const torch::Tensor& geomBuffer,
                               const int R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                               const int raw_params, const torch::Tensor& dL_dcolor_view, const ShAdamStep& sh_adam,
                               const std::vector<torch::Tensor>& view_stats, const GeomAdamStep& geom_adam);

// gsr_sh_grad_from_views (include/gsr.h): the [P,M,3] SH gradient of a keyframe batch from the gathered
// [n_views,P,3] dL_dcolor_view tensors and the [n_views,3] camera centres; scale = 1/n_views for the batch mean
torch::Tensor shGradFromViews(const torch::Tensor& means3D, const torch::Tensor& campos_views,
                              const torch::Tensor& dL_dcolor_views, const int degree, const int M, const float scale);

// gsr_sh_adam_from_views: the same rebuild with this step's Adam update of `sh` [P,16,3] applied in place (no gradient tensor)
void shAdamFromViews(const torch::Tensor& means3D, const torch::Tensor& campos_views, const torch::Tensor& dL_dcolor_views,
                     const int degree, const float scale, torch::Tensor& sh, const ShAdamStep& sh_adam);

// The PACKED form of the exchange (include/gsr.h: gsr_pack_color_view): a view's [P,3] colour gradient as a message of the rows the
// view SEES (mask + rows + camera centre).  lastVisibleCount(): radii > 0 in this thread's last RasterizeGaussiansCUDA (= the
// rows a message of that view holds); packedViewWords(): int32 words of a message with room for `capacity` rows (a multiple of 4,
// the same on every rank); packColorView(): writes `message` (int32, >= packedViewWords words; `scratch` is a uint8 tensor grown as
// needed) on the CURRENT stream; the two consumers are shGradFromViews / shAdamFromViews on n_views messages msg_stride words apart.
int lastVisibleCount();
int64_t packedViewWords(int64_t P, int64_t capacity);
// throws unless each of the n_views gathered messages says "P rows, this capacity, nothing dropped" (gsr_check_packed_views; WAITS
// for the current stream: tests, the first steps of a session, debugging runs)
void checkPackedViews(const torch::Tensor& messages, int64_t msg_stride, int64_t n_views, int64_t P, int64_t capacity);
void packColorView(const torch::Tensor& dL_dcolor_view, const torch::Tensor& campos, int64_t capacity, torch::Tensor& message,
                   torch::Tensor& scratch);
// gsr_pack_view_plan: the mask and prefix sections of a view's message from the forward pass's radii (CURRENT stream); the
// backward pass writes rows and header itself when ShAdamStep::packed_view names the message
void packViewPlan(const torch::Tensor& radii, int64_t capacity, torch::Tensor& message, torch::Tensor& scratch);
torch::Tensor shGradFromPackedViews(const torch::Tensor& means3D, const torch::Tensor& messages, int64_t msg_stride, int64_t n_views,
                                    const int degree, const int M, const float scale);
void shAdamFromPackedViews(const torch::Tensor& means3D, const torch::Tensor& messages, int64_t msg_stride, int64_t n_views,
                           const int degree, const float scale, torch::Tensor& sh, const ShAdamStep& sh_adam);

// gsr_sh_adam_flush: lazy mode -- every row of `sh` takes the zero-gradient steps it is behind, up to sh_adam.step = the number
// of Adam steps the tensor has taken (sh_adam.lr / lr_tail belong to that step)
void shAdamFlush(torch::Tensor& sh, const ShAdamStep& sh_adam);
// gsr_sh_adam_lazy_slice: the data-parallel step's rotating catch-up of the rows no view lights -- after the last
// shAdamFromViews() range of the step (up to sh_adam.step), or `ahead` of them (up to step - 1: what the rasterizer's backward
// does by itself in the view-factored mode, so TrainStep never calls this)
void shAdamLazySlice(torch::Tensor& sh, const ShAdamStep& sh_adam, bool ahead = false);

// gsr_adam_step_multi: one Adam step (gsr_adam_step arithmetic) of several tensors in ONE launch
struct AdamMultiEntry {
	torch::Tensor param, grad, exp_avg, exp_avg_sq;   // contiguous float32, one size
	double lr = 0.0;
	int step = 0;
	float grad_scale = 1.0f;   // multiplies the gradient as it is read
};
void adamStepMulti(const std::vector<AdamMultiEntry>& entries, double beta1, double beta2, double eps);

torch::Tensor markVisible(torch::Tensor& means3D, torch::Tensor& viewmatrix, torch::Tensor& projmatrix);
