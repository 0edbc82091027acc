// kernels.h -- parameter blocks and host-side launchers of every device stage.
#pragma once
#include "state.h"

#include <math.h>

namespace gsr {

// The scalars one Adam step needs, derived ONCE on the host from torch::optim::AdamOptions-style double hyper-parameters
// (torch forms bias_correction = 1 - beta^step, 1 - beta and lr / bias_correction1 in double and hands fp32 tensors the
// rounded results): m = b1 m + omb1 g; v = b2 v + omb2 g g; p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps).
struct AdamScalars {
	float step_size, step_size_tail, b1, b2, omb1, omb2, eps, inv_sqrt_bc2;
};
static inline AdamScalars adam_scalars(double lr, double lr_tail, double beta1, double beta2, double eps, int step)
{
	const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
	AdamScalars a;
	a.step_size = (float)(lr / bc1);
	a.step_size_tail = (float)(lr_tail / bc1);
	a.b1 = (float)beta1;
	a.b2 = (float)beta2;
	a.omb1 = (float)(1.0 - beta1);
	a.omb2 = (float)(1.0 - beta2);
	a.eps = (float)eps;
	a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
	return a;
}

// Lazy Adam for the SH rows of culled Gaussians (gsr_sh_adam_lazy; the mechanism is described in shrows.h)
constexpr int LAZY_WINDOW_MAX = 32;   // == GSR_SH_LAZY_WINDOW
constexpr int LAZY_BLOCK_ROWS = 64;   // granularity of the rotating slice: row block b is due at steps s with s % window == b % window
struct LazyAdamTable {                // [k]: the scalars of Adam step (step - k)
	float step_size[LAZY_WINDOW_MAX], step_size_tail[LAZY_WINDOW_MAX], inv_sqrt_bc2[LAZY_WINDOW_MAX];
	float b1, b2, omb1, omb2, eps;
};
struct LazyAdam {
	float* param;        // [P][48]
	float* exp_avg;
	float* exp_avg_sq;
	int* row_step;       // [P]; null = off
	int step;            // the Adam step the tensor is taking (forward / backward) or has taken (flush)
	int window;
	LazyAdamTable t;
};

struct PreprocessParams {
	int P, D, M;
	const float* means3D;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* opacities;
	const float* shs;
	const float* cov3D_precomp;
	const float* colors_precomp;
	const float* view;    // [16] device
	const float* proj;    // [16] device
	const float* campos;  // [3]  device
	int W, H;
	float tan_fovx, tan_fovy, focal_x, focal_y;
	int grid_x, grid_y;
	int* radii_out;  // caller's radii (nullable)
	int raw_params;  // GSR_RAW_* mask: activations applied in-kernel
	uint2* ranges;   // [tiles] per-tile instance ranges: zeroed here (identifyTileRanges fills only the tiles that have instances)
	int tiles;
	LazyAdam lazy;   // row_step != null: visible rows that lag behind (step - 1) are brought up to date before their SH evaluation
};
int launch_preprocess_fwd(const PreprocessParams& p, const GeometryState& g, hipStream_t stream);
int launch_check_frustum(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t stream);

// the instance emission's waves own EMIT_SEED_STRIDE consecutive slots each; the offset scan leaves, for every such window, the
// depth rank of the Gaussian that holds its first slot (GeometryState::sort_keys_b, P entries: windows beyond that search)
constexpr uint32_t EMIT_SEED_STRIDE = 256;
int launch_emit_instances(int P, int R, const GeometryState& g, int grid_x, uint32_t* keys, uint32_t* vals, uint8_t* touched,
                          hipStream_t stream, int cull = 0, bool seeded = false, uint32_t* hist = nullptr, int hist_bits = 0);
int launch_tile_ranges(int R, const uint32_t* tile_keys, uint2* ranges, hipStream_t stream, const uint32_t* n_dev = nullptr);

struct BlendFwdParams {
	const uint2* ranges;
	const uint32_t* point_list;
	const float4* rec;
	const float* bg;
	float* final_T;
	uint32_t* n_contrib;
	float* out_color;
	uint8_t* contrib;       // [4][contrib_stride]: plane of quad q, byte per list entry (state.h)
	size_t contrib_stride;
	int W, H, grid_x, tiles;
	TileDeal deal;          // blend.h: the workgroup -> XCD deal of the tiles
};
int launch_blend_fwd(const BlendFwdParams& p, hipStream_t stream);

struct BlendBwdParams {
	const uint2* ranges;
	const uint32_t* point_list;
	const float4* rec;
	const float* bg;
	const float* final_T;
	const uint32_t* n_contrib;
	const float* dL_dpix;   // [3,H,W]
	float* partials;        // [R][12] per-instance gradient slots (blend.h); only slots flagged in `touched` are meaningful
	uint8_t* touched;       // [R] zeroed by the caller; set to 1 for every slot written
	const uint8_t* contrib; // [4][contrib_stride] the forward blend's per-quad contribution flags
	size_t contrib_stride;
	int W, H, grid_x, tiles;
	TileDeal deal;          // blend.h: the workgroup -> XCD deal of the tiles
};
int launch_blend_bwd(const BlendBwdParams& p, hipStream_t stream);

// Fused Adam steps of the four per-Gaussian geometry tensors (gsr_geom_adam): on == 0 = off
struct GeomAdamTensor {
	float* param;
	float* exp_avg;
	float* exp_avg_sq;
	AdamScalars s;
};
struct GeomAdam {
	int on;
	GeomAdamTensor xyz, opacity, scaling, rotation;
};

struct PreprocessBwdParams {
	int P, D, M;
	const float* means3D;
	const int* radii;
	const float* shs;
	const uint8_t* clamped;
	const float* scales;
	const float* rotations;
	float scale_modifier;
	const float* cov3D;     // cov3D_precomp, or null: recomputed from scales / rotations (compute_cov3D)
	const float* view;      // [16] device
	const float* proj;      // [16] device
	const float* campos;    // [3] device
	float focal_x, focal_y, tan_fovx, tan_fovy;
	const uint32_t* tiles_touched;   // [P] length of each Gaussian's run of instance slots (0 = culled)
	float* partials;          // [R][12] per-instance gradient slots written by the backward blend (blend.h); the totals of the
	                          // long runs are folded into their first slots (long_run_sums_kernel)
	uint8_t* touched;         // [R + 64] 1 where a slot was written
	const uint32_t* long_runs;       // ids of the Gaussians with more than LONG_RUN slots, LONG_LISTS sub-lists (the offset scan)
	const uint32_t* long_counts;     // entries per sub-list (device)
	uint32_t long_capacity;
	float half_w, half_h;     // W/2, H/2: the ndc -> pixel factors of dL_dmean2D (backward.cu:460-461)
	const float4* rec;        // [3P] blend records (activated opacity for the raw-parameter chain rule)
	float* dL_dmean2D;        // [P,3]  unpacked here (x, y, 0); nullable
	float* dL_dconic;         // [P,4]  nullable
	float* dL_dopacity;       // [P]
	float* dL_dcolor;         // [P,3]
	float* dL_dmean3D;        // [P,3]  (with geom.on: scratch between the two kernels, not an output)
	float* dL_dcov3D;         // [P,6]  nullable
	float* dL_dsh;            // [P,M,3] nullable
	float* dL_dcolor_view;    // [P,3] nullable: view-factored mode (gsr.h) -- the clamp-masked colour gradient INSTEAD of dL_dsh
	float* dL_dscale;         // [P,3] nullable
	float* dL_drot;           // [P,4] nullable
	int raw_params;           // GSR_RAW_* mask: outputs are gradients of the raw parameters
	// densification statistics of this view fused in (gsr_backward_args.stat_*): null = off
	float* stat_accum;
	float* stat_denom;
	float* stat_max_radii;
	// fused Adam step of the SH tensor (gsr_backward_args.sh_adam): dL_dsh never leaves the LDS rows; null exp_avg = off
	float* adam_param;        // == shs, written
	float* adam_exp_avg;
	float* adam_exp_avg_sq;
	AdamScalars adam;
	int adam_skip_culled;     // the culled Gaussians' rows take this step elsewhere (gsr_backward: side stream) or later (lazy)
	int* lazy_row_step;       // lazy mode: row_step[i] = lazy_step for the rows updated here (the visible ones); null = off
	int lazy_step;
	// the slot flags are left cleared for the next backward pass by sh_bwd_rows_kernel (null: the caller clears them itself)
	uint8_t* touched_clear;
	uint32_t touched_clear_bytes;
	// optimizer-in-backward for xyz / opacity / scaling / rotation (gsr_backward_args.geom_adam): their gradients are not written
	GeomAdam geom;
	// view-factored mode (gsr_backward_args.color_view_ready_stream): dL_dcolor_view is complete when preprocess_bwd_kernel has
	// run -- notify_stream is made to wait for exactly that point (notify_event recorded between the two kernels), so that a
	// gather issued on it overlaps sh_bwd_rows_kernel.  Host-side only; null = off.
	void* notify_stream;
	void* notify_event;
	// view-factored mode, packed (gsr_backward_args.packed_view): the view's message with its prefix and mask sections filled by
	// gsr_pack_view_plan -- preprocess_bwd_kernel writes the seen rows and the header next to dL_dcolor_view; null = off
	uint32_t* packed_msg;
	int packed_capacity;      // rows the message has room for
};
int launch_preprocess_bwd(const PreprocessBwdParams& p, hipStream_t stream);
// does the backward preprocess take the two-kernel path of the reference's SH layout (preprocess_bwd_kernel<true> +
// sh_bwd_rows_kernel)?  (aligned 48-float rows, a row consumer: gradient rows out, the factored colour gradient, or the fused step)
static inline bool sh_rows_path(const float* shs, int M, int D, bool factored, bool adam, const float* dL_dsh)
{
	return shs && 3 * M == 48 && (reinterpret_cast<uintptr_t>(shs) & 15) == 0 && D >= 0 && D <= 3 &&
	       (factored || adam || (dL_dsh && (reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0));
}
// Packed colour-gradient messages (include/gsr.h: gsr_pack_color_view): layout helpers shared by the packer and the decoder
constexpr int PACK_HEADER = 8;
static inline size_t pack_groups(int P) { return ((size_t)P + 63) / 64; }
static inline size_t pack_prefix_words(int P) { return (pack_groups(P) + 3) & ~(size_t)3; }
static inline size_t pack_mask_words(int P) { return (2 * pack_groups(P) + 3) & ~(size_t)3; }
static inline size_t packed_view_words(int P, int capacity) { return PACK_HEADER + pack_prefix_words(P) + pack_mask_words(P) + 3 * (size_t)capacity + 4; }
int launch_pack_color_view(int P, const float* view, const float* campos, int capacity, uint32_t* msg, uint32_t* scratch, hipStream_t stream);
// the capacity-independent part of a view's message from the forward pass's radii (seen = radii > 0): masks + prefix
int launch_pack_view_plan(int P, const int* radii, uint32_t* msg, uint32_t* scratch, hipStream_t stream);
struct PackedViews {   // n_views messages, `stride` words apart; null msgs = the dense [n_views, P, 3] form
	const uint32_t* msgs = nullptr;
	long long stride = 0;
};
// dL_dsh from the per-view colour gradients of a keyframe batch (gsr_sh_grad_from_views)
struct RowAdam;   // shrows.h: Adam state + scalars of the fused row update (null = write the gradient rows)
int launch_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                              long long campos_stride, const float* dL_dcolor_views, long long view_stride, float scale,
                              float* dL_dsh, const RowAdam* adam, hipStream_t stream, const LazyAdam* lazy = nullptr,
                              PackedViews packed = PackedViews());

// this step's Adam update of the [P,16,3] SH rows of the CULLED Gaussians (radii <= 0; zero gradient): gsr_backward, side stream
int launch_sh_adam_culled(int P, const int* radii, const RowAdam& adam, hipStream_t stream, int max_blocks);

// lazy mode: the zero-gradient steps rows are behind (preprocess_bwd.hip: sh_adam_lazy_kernel).  mode 0: this step's slice,
// rows with radii <= 0, up to a.step (the fused backward); 1: every row of every block (gsr_sh_adam_flush); 2: this step's slice,
// every row (gsr_sh_adam_lazy_slice); 3: the slice of period window - 1, every row, up to a.step - 1 (the data-parallel step,
// next to the backward blend).  radii: mode 0 only.
int launch_sh_adam_lazy(int P, const int* radii, const LazyAdam& a, hipStream_t stream, int mode);

// simple-knn
size_t knn_scratch_bytes(int P);
int launch_knn(int P, const float* points, float* meanDists, char* scratch, hipStream_t stream);
// the first half of it on its own: the points' order along the Z-order curve of their bounding box (*order_out: n ids inside
// `scratch`, knn_scratch_bytes(n) bytes; *points_buf: 3 n floats inside scratch the caller may fill and pass as `points`)
int launch_morton_order(int n, const float* points, uint32_t** order_out, char* scratch, hipStream_t stream);
float* morton_points_buffer(int n, char* scratch);

// computeCov3D, forward.cu:118-152 (M = S*R with S diagonal: M[c][r] = s_r * R[c][r]; Sigma = transpose(M) * M), with the
// activations of raw_params applied first (getScalingActivation / getRotationActivation, gaussian_model.cpp:48-56).
// ONE function for the forward preprocess, which needs Sigma for the projection, and for the backward preprocess, which
// needs it again for computeCov2DCUDA (backward.cu:144-274): the reference stores 24 bytes per Gaussian in between
// (geomState.cov3D), here the backward pass recomputes them from the scale and rotation it loads anyway -- both translation
// units are compiled with -ffp-contract=off, so the two evaluations are the same bits.
__device__ __forceinline__ void compute_cov3D(const float* __restrict__ scales, const float* __restrict__ rotations, size_t idx,
                                              float scale_modifier, int raw_params, float c3[6])
{
	float sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
	if (raw_params & GSR_RAW_SCALING) {   // getScalingActivation, gaussian_model.cpp:48-51
		sx = expf(sx);
		sy = expf(sy);
		sz = expf(sz);
	}
	const float s0 = scale_modifier * sx, s1 = scale_modifier * sy, s2 = scale_modifier * sz;
	float4 q = reinterpret_cast<const float4*>(rotations)[idx];
	if (raw_params & GSR_RAW_ROTATION) {  // getRotationActivation: F::normalize (eps 1e-12), :53-56
		const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
		q.x = q.x / qn;
		q.y = q.y / qn;
		q.z = q.z / qn;
		q.w = q.w / qn;
	}
	const float r = q.x, x = q.y, y = q.z, z = q.w;
	// R[c][r] (glm column-major) exactly as written at forward.cu:135-139
	const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
	const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
	const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
	const float M00 = s0 * R00, M01 = s1 * R01, M02 = s2 * R02;
	const float M10 = s0 * R10, M11 = s1 * R11, M12 = s2 * R12;
	const float M20 = s0 * R20, M21 = s1 * R21, M22 = s2 * R22;
	// Sigma = transpose(M) * M:  Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]
	c3[0] = M00 * M00 + M01 * M01 + M02 * M02;
	c3[1] = M10 * M00 + M11 * M01 + M12 * M02;
	c3[2] = M20 * M00 + M21 * M01 + M22 * M02;
	c3[3] = M10 * M10 + M11 * M11 + M12 * M12;
	c3[4] = M20 * M10 + M21 * M11 + M22 * M12;
	c3[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

}  // namespace gsr
