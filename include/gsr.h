/*
 * gsr.h -- C-ABI of the MI355X-native Gaussian-splatting rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary of the hot path: plain pointers and sizes, no torch
 * types.  Every entry point replaces one L0 interface of the reference
 * (HuajianUP/Photo-SLAM); the LibTorch wrappers RasterizeGaussiansCUDA /
 * RasterizeGaussiansBackwardCUDA / markVisible / distCUDA2 sit directly on top
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * All pointers are DEVICE pointers (gfx950 HBM) unless stated; fp32 throughout.
 * A null pointer stands for an absent optional exactly as in the reference, where
 * an empty tensor yields data_ptr()==nullptr (src/gaussian_rasterizer.cpp:209-219,
 * cuda_rasterizer/forward.cu:205,241).
 * The library owns no device memory: scratch comes from the caller through
 * gsr_alloc_fn (the reference's std::function<char*(size_t)> resize callbacks,
 * cuda_rasterizer/rasterizer.h:36-38, src/rasterize_points.cu:28-34).
 * All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null
 * stream).  Functions return GSR_OK or a negative gsr_status; they never throw.
 */
#ifndef GSR_H
#define GSR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum gsr_status {
	GSR_OK = 0,
	GSR_ERR_INVALID_ARG = -1,   /* bad shape / forbidden null / inconsistent optional combination */
	GSR_ERR_ALLOC = -2,         /* a gsr_alloc_fn returned NULL */
	GSR_ERR_HIP = -3,           /* a HIP runtime call or kernel launch failed (see gsr_last_hip_error) */
	GSR_ERR_UNSUPPORTED = -4    /* e.g. tile grid larger than 65535 in one dimension */
} gsr_status;

/* Resize-and-return-pointer callback; must return a device pointer to at least
 * `bytes` bytes, 16-byte aligned (torch allocations are 512-byte aligned).
 * Replaces std::function<char*(size_t)> of Rasterizer::forward
 * (cuda_rasterizer/rasterizer.h:36-38). */
typedef char* (*gsr_alloc_fn)(void* ctx, size_t bytes);

/* Extension of the extension below: LAZY Adam steps for the SH rows of culled Gaussians (NULL in gsr_sh_adam = every row takes
 * every step when it happens).  A Gaussian the view culls gets a zero SH gradient, and a zero-gradient Adam step of a row --
 * m <- b1 m, v <- b2 v, p <- p - step_size m / (sqrt(v) / sqrt(1 - b2^t) + eps) -- depends on nothing but the row and the
 * step's scalars.  With row_step set the library takes such steps LATER, several at a time with the row in registers: the
 * same arithmetic in the same order (results bit-identical to the eager update, tests/test_lazy_sh_adam.py), one HBM round
 * trip of the row's 1152 bytes instead of one per step.  row_step[i] = number of Adam steps row i has taken.  A row is
 * brought up to date
 *   - by gsr_forward when it becomes visible, BEFORE its coefficients are evaluated (pass the same gsr_sh_adam to
 *     gsr_forward_args.sh_adam and gsr_backward_args.sh_adam of a step);
 *   - by gsr_backward for a rotating 1/window of the row blocks per step, so that no row ever lags by more than `window`
 *     steps (the past learning rates the caller has to remember);
 *   - by gsr_sh_adam_flush for all rows: REQUIRED before anything else reads or writes the tensor or its moments (densify /
 *     prune, save, a dense optimizer step, another exchange mode).
 * Contract: when the first lazy step `step` is taken every row_step[i] == step - 1; from then on every step's forward and
 * backward get the struct with the same `window` until a flush. */
#define GSR_SH_LAZY_WINDOW 32
typedef struct gsr_sh_adam_lazy {
	int* row_step;                          /* [P] device ints, read and written */
	int window;                             /* 2 .. GSR_SH_LAZY_WINDOW */
	double lr_past[GSR_SH_LAZY_WINDOW];     /* [k-1] = lr of Adam step (step - k), k = 1 .. window-1 (entries of steps < 1 unused) */
	double lr_tail_past[GSR_SH_LAZY_WINDOW];
} gsr_sh_adam_lazy;

/* Extension: Adam state of the [P,16,3] SH tensor for the fused update inside gsr_backward (see
 * gsr_backward_args.sh_adam).  torch::optim::Adam semantics as gsr_adam_step; the first 3 floats of a row
 * (features_dc) use lr, the other 45 (features_rest) lr_tail. */
typedef struct gsr_sh_adam {
	float* param;                /* [P,16,3]: the SH tensor itself, UPDATED IN PLACE; gsr_backward requires param == shs (the
	                                const input pointer of the reference's parameter list stays const: the write goes
	                                through this one) */
	float* exp_avg;              /* [P,16,3] */
	float* exp_avg_sq;           /* [P,16,3] */
	double lr, lr_tail, beta1, beta2, eps;   /* double like torch::optim::AdamOptions: the bias corrections 1 - beta^step are
	                                            formed in double as torch does (0.999f instead of 0.999 is 1e-5 of the step) */
	int step;                    /* >= 1: the step being taken (bias correction) */
	const gsr_sh_adam_lazy* lazy; /* NULL = eager: every row takes the step in gsr_backward */
	/* Where the optimizer work that gsr_forward / gsr_backward fork next to their own kernels runs.  Zero in every field = the
	 * measured-best arrangement on MI355X (DESIGN.md sections 5, 9.1) -- a zero-initialised struct is the fast one.  None of
	 * them changes a result.  The environment variables named here OVERRIDE the fields when set (A/B handles of the bench
	 * sessions); a process that never sets them is governed by the fields alone. */
	int no_side_stream;          /* 1: no second stream at all -- the lazy rows' slice and the culled rows of the eager step run on
	                                the caller's stream (GSR_SH_ADAM_SIDE_STREAM=0 / 1) */
	int lazy_slice_late;         /* 1: this step's slice of the lazy rows is forked behind the backward blend instead of next to it
	                                (GSR_LAZY_SLICE_EARLY=0 / 1; measured: next to the blend C3 1.625 -> 1.613 ms) */
	int side_blocks;             /* eager mode: workgroups of the culled rows' kernel on the second stream; 0 = 256 (measured best
	                                of 64 / 256 / 1024 / 2048; GSR_SH_ADAM_SIDE_BLOCKS) */
} gsr_sh_adam;

/* Extension: optimizer-in-backward for the four per-Gaussian geometry tensors (see gsr_backward_args.geom_adam).  One entry per
 * tensor: torch::optim::Adam semantics as gsr_adam_step, each tensor with its own learning rate and step counter. */
typedef struct gsr_adam_tensor {
	float* param;                /* UPDATED IN PLACE */
	float* exp_avg;
	float* exp_avg_sq;
	double lr;
	int step;                    /* >= 1: the step being taken */
} gsr_adam_tensor;
typedef struct gsr_geom_adam {
	gsr_adam_tensor xyz;         /* [P,3]; param must be means3D */
	gsr_adam_tensor opacity;     /* [P]   the raw opacity (logits) */
	gsr_adam_tensor scaling;     /* [P,3]; param must be scales (log-scales) */
	gsr_adam_tensor rotation;    /* [P,4]; param must be rotations (unnormalised quaternions) */
	double beta1, beta2, eps;
} gsr_geom_adam;

/* Rasterizer::forward parameter list, cuda_rasterizer/rasterizer.h:35-59, 1:1. */
typedef struct gsr_forward_args {
	int P, D, M;                 /* #Gaussians, active SH degree, SH coeffs per channel stored */
	const float* background;     /* [3] */
	int width, height;
	const float* means3D;        /* [P,3] */
	const float* shs;            /* [P,M,3] or NULL */
	const float* colors_precomp; /* [P,3] or NULL (exactly one of shs / colors_precomp) */
	const float* opacities;      /* [P] */
	const float* scales;         /* [P,3] or NULL */
	float scale_modifier;
	const float* rotations;      /* [P,4] (r,x,y,z), NOT normalised in-kernel (forward.cu:127) */
	const float* cov3D_precomp;  /* [P,6] or NULL (exactly one of scales+rotations / cov3D_precomp) */
	const float* viewmatrix;     /* [16] = W2C^T row-major, i.e. element (r,c) at [4c+r] */
	const float* projmatrix;     /* [16] = (Proj W2C)^T */
	const float* cam_pos;        /* [3] */
	float tan_fovx, tan_fovy;
	int prefiltered;
	float* out_color;            /* [3,H,W] written for every pixel */
	int* radii;                  /* [P] or NULL */
	/* Extension (0 = the reference contract: activated inputs).  Bit mask of GSR_RAW_*: the corresponding input
	 * holds the model's RAW parameter and the activation of GaussianModel (src/gaussian_model.cpp:48-62) is applied
	 * in-kernel: sigmoid(opacity), exp(scaling), normalize(rotation).  Saves the ~13 elementwise ATen launches
	 * (forward + autograd) GaussianRenderer::render otherwise spends per step. */
	int raw_params;
	/* Extension (NULL = the reference contract; consulted only when sh_adam->lazy is set, see gsr_sh_adam_lazy): the SH
	 * rows of visible Gaussians that lag behind (sh_adam->step - 1) take their missed zero-gradient Adam steps before their
	 * coefficients are evaluated -- the forward pass then WRITES sh_adam->param (== shs), the moments and row_step. */
	const gsr_sh_adam* sh_adam;
} gsr_forward_args;

#define GSR_RAW_OPACITY 1   /* opacities are logits */
#define GSR_RAW_SCALING 2   /* scales are log-scales */
#define GSR_RAW_ROTATION 4  /* rotations are unnormalised quaternions (normalised with eps 1e-12 like F::normalize) */
/* One more bit of gsr_forward_args.raw_params (ignored by gsr_backward), not about the inputs: the reference lists a Gaussian in
 * EVERY tile of the bounding square of its 3-sigma radius (duplicateWithKeys, rasterizer_impl.cu:70-111); most of those
 * instances blend into no pixel (alpha < 1/255 over the whole tile: 78 % of them at 2 M Gaussians @ 1080p).  With this bit the
 * instances of such tiles -- decided by the same conservative bound the blend kernels apply per 8x8 quad, on the tile's 16x16
 * rectangle -- are dropped in front of the tile sort.  The image and every gradient are the SAME, bit for bit (the dropped
 * pairs are pairs the reference `continue`s over at every pixel); what changes is internal: the sorted instance list is
 * shorter (num_rendered still counts the rectangles: it sizes the buffers), and n_contrib counts positions of the shorter
 * list.  0 = the reference's lists, bit for bit.  Measured: 26-40 % of the instances go, the sort and
 * the blend kernels gain what the test costs in the emission -- both hosts leave it off unless GSR_CULL_EMPTY_TILES=1. */
#define GSR_CULL_EMPTY_TILES 8
/* ... and one for introspection (ignored by gsr_backward): the forward pass ALSO leaves the 3-D covariances it computed in the
 * geometry buffer (24 bytes per visible Gaussian; the reference's geomState.cov3D).  Nothing in the library reads them back --
 * gsr_backward recomputes them from scales / rotations, the same arithmetic and the same bits -- so by default they are not
 * written; the test-suite's view into the buffer (tests/dev) asks for them. */
#define GSR_STORE_COV3D 16
/* ... and two that choose how the per-tile lists are built (ignored by gsr_backward; neither = the library decides by the size
 * of the view; the lists, the image and the gradients are the same either way, bit for bit).  DEPTH_FIRST: the Gaussians are
 * sorted by depth once (nine launches whatever the size), their instances emitted in that order, and the stable tile sort carries
 * the order into the tiles -- the least work per instance, the arrangement for large views.  TILE_FIRST: the visible Gaussians are
 * only compacted (ascending id) and every tile's list is sorted by depth on its own behind the tile sort, one workgroup per tile
 * in LDS -- eight launches fewer, the arrangement for the small and mid-size views a SLAM session starts with. */
#define GSR_BINNING_DEPTH_FIRST 32
#define GSR_BINNING_TILE_FIRST 64

/* Rasterizer::forward, cuda_rasterizer/rasterizer_impl.cu:198-336.
 * Fills out_color and radii, returns the number of (tile, Gaussian) instances in
 * *num_rendered.  The three scratch buffers are opaque and must be handed unchanged
 * to gsr_backward.  One host synchronisation (to size the binning buffer), as the
 * reference (rasterizer_impl.cu:281).  P == 0 is a valid no-op that leaves
 * out_color untouched (src/rasterize_points.cu:81). */
int gsr_forward(const gsr_forward_args* args,
                gsr_alloc_fn geometryBuffer, void* geometry_ctx,
                gsr_alloc_fn binningBuffer, void* binning_ctx,
                gsr_alloc_fn imageBuffer, void* image_ctx,
                void* stream, int* num_rendered);


/* Rasterizer::backward parameter list, cuda_rasterizer/rasterizer.h:61-91. */
typedef struct gsr_backward_args {
	int P, D, M, R;
	const float* background;
	int width, height;
	const float* means3D;
	const float* shs;
	const float* colors_precomp;
	const float* scales;
	float scale_modifier;
	const float* rotations;
	const float* cov3D_precomp;
	const float* viewmatrix;
	const float* projmatrix;
	const float* campos;
	float tan_fovx, tan_fovy;
	const int* radii;            /* [P] or NULL (then the copy inside geom_buffer is used) */
	char* geom_buffer;
	char* binning_buffer;
	char* image_buffer;
	const float* dL_dpix;        /* [3,H,W] */
	float* dL_dmean2D;           /* [P,3]  (.z stays 0); NULL is accepted (a caller that fuses the densification statistics,
	                                stat_* below, has no other use for it) */
	float* dL_dconic;            /* [P,4]  the reference's [P,2,2] (.z = 0); internal to the reference's wrapper
	                                (rasterize_points.cu:152), so NULL is accepted */
	float* dL_dopacity;          /* [P]   */
	float* dL_dcolor;            /* [P,3] */
	float* dL_dmean3D;           /* [P,3] */
	float* dL_dcov3D;            /* [P,6]; NULL is accepted when cov3D_precomp is NULL (the gradient continues into scales and
	                                rotations inside the kernel) */
	float* dL_dsh;               /* [P,M,3] or NULL when shs is NULL */
	float* dL_dscale;            /* [P,3] or NULL when scales is NULL */
	float* dL_drot;              /* [P,4] or NULL when scales is NULL */
	/* Same mask as gsr_forward_args.raw_params (must match the forward call): dL_dopacity / dL_dscale / dL_drot are
	 * then gradients w.r.t. the RAW parameters (chain rule through sigmoid / exp / normalize applied in-kernel). */
	int raw_params;
	/* Extension for keyframe-batch data parallelism (NULL = the reference contract).  [P,3]: when set, dL_dsh is NOT
	 * written (and may be NULL); instead the gradient w.r.t. the SH colour BEFORE the clamp -- dL_dcolor with the
	 * clamped channels zeroed (backward.cu:41-48), zeros for culled Gaussians -- is written here (a VISIBLE Gaussian whose masked
	 * gradient is all zero carries -0.0f in channel 0: numerically nothing, a visibility marker for the lazy rows of
	 * gsr_sh_adam_from_views).  dL_dsh of one view is
	 * basis(dir) x this vector; gsr_sh_grad_from_views rebuilds it for all views after the exchange.  The SH term of
	 * dL_dmean3D is computed as usual. */
	float* dL_dcolor_view;
	/* Extension, optimizer-in-backward for the SH tensor (NULL = the reference contract).  When set, dL_dsh is NOT written
	 * (and may be NULL): the kernel that produces the gradient rows applies this step's Adam update to sh_adam->param (which
	 * must be the same tensor as shs) IN PLACE and to the two moment tensors, for every Gaussian (culled ones with a zero
	 * gradient, as a dense optimizer does) -- the 192 B/Gaussian gradient row never round-trips through HBM.  The rows of
	 * the Gaussians this view culls do not depend on the backward pass at all (zero gradient): gsr_backward updates them on
	 * a second HIP stream it owns, concurrently with the VALU-bound backward blend that leaves HBM nearly idle, and joins
	 * that stream before it returns (environment GSR_SH_ADAM_SIDE_STREAM=0: everything on the caller's stream).  Only for
	 * 16-byte aligned [P,16,3] tensors (GSR_ERR_UNSUPPORTED otherwise).  Together with dL_dcolor_view only in the lazy form
	 * (sh_adam->lazy, window >= 3), and then with another meaning: no step is fused (it follows the exchange:
	 * gsr_sh_adam_from_views); backward runs, on its second stream next to the blend kernel, the rotating catch-up that
	 * bounds the lag of the rows no view lights -- the row blocks b with b % (window - 1) == step % (window - 1) take the
	 * zero-gradient steps they are behind up to step - 1 (what happens AT step is decided by the exchange). */
	const gsr_sh_adam* sh_adam;
	/* Extension (all three or none; NULL = the reference contract): the densification statistics of this view, updated by
	 * the kernel that holds dL_dmean2D in registers instead of a separate pass (gsr_densify_stats, same arithmetic): for
	 * radii > 0: stat_grad_accum += |dL_dmean2D.xy|, stat_denom += 1, stat_max_radii = max(., radii).  [P] floats each. */
	float* stat_grad_accum;
	float* stat_denom;
	float* stat_max_radii;
	/* Extension, optimizer-in-backward for xyz / opacity / scaling / rotation (NULL = the reference contract).  The kernels that
	 * hold these four gradients in registers apply this step's Adam update instead of writing them: dL_dopacity, dL_dscale and
	 * dL_drot are NOT written (and may be NULL), dL_dmean3D is still required but only as scratch between two kernels.  Saves
	 * the 88 B per Gaussian gradient round trip and four optimizer launches.  Needs raw_params == GSR_RAW_OPACITY |
	 * GSR_RAW_SCALING | GSR_RAW_ROTATION (the gradients must be those of the tensors being stepped), scales + rotations (no
	 * cov3D_precomp) and the 16-byte aligned [P,16,3] SH layout (GSR_ERR_UNSUPPORTED otherwise); every Gaussian steps, culled
	 * ones with a zero gradient as a dense optimizer does. */
	const gsr_geom_adam* geom_adam;
	/* Extension for the view-factored exchange (NULL = off; a hipStream_t, consulted only with dL_dcolor_view): dL_dcolor_view is
	 * complete before the last kernel of the backward pass (which only READS it, for the view-direction term of dL_dmean3D).
	 * gsr_backward makes this stream wait for exactly that point, so that a gather the caller issues on it -- the RCCL
	 * all-gather of the exchange -- overlaps that last kernel instead of following the whole pass.  The stream must not be the
	 * one gsr_backward is called with. */
	void* color_view_ready_stream;
	/* Extension for the PACKED view-factored exchange (NULL = off; consulted only with dL_dcolor_view): a message whose prefix and
	 * mask sections gsr_pack_view_plan has filled from this view's radii.  The backward pass writes the seen rows and the header
	 * into it next to dL_dcolor_view -- the message is complete at the same point as the dense view (color_view_ready_stream),
	 * with no pack launch between the backward pass and the gather.  packed_capacity_rows = the rows the buffer has room for
	 * (>= this view's visible count, e.g. P rounded up to 4: the ranks then send the first gsr_packed_view_words(P, max_v K_v)
	 * words).  The same bits as gsr_pack_color_view(P, dL_dcolor_view, campos, packed_capacity_rows, ...). */
	uint32_t* packed_view;
	int packed_capacity_rows;
} gsr_backward_args;

/* Rasterizer::backward, cuda_rasterizer/rasterizer_impl.cu:340-433.
 * Unlike the reference the gradient arrays need NOT be zero-filled by the caller:
 * every element of every non-null output is written (zeros for culled Gaussians),
 * which removes the reference's 300 B/Gaussian torch::zeros pass
 * (src/rasterize_points.cu:149-157).  No host synchronisation. */
int gsr_backward(const gsr_backward_args* args, void* stream);

/* Lazy SH Adam (gsr_sh_adam_lazy): every row of the [P,16,3] tensor takes the zero-gradient steps it is behind, up to and
 * including adam->step = the number of Adam steps the tensor has taken (adam->lr / lr_tail belong to that step, lr_past[k-1] to
 * step - k); afterwards row_step[i] == adam->step for all i and tensor and moments are what the eager update leaves. */
int gsr_sh_adam_flush(int P, const gsr_sh_adam* adam, void* stream);

/* The SH gradient of a keyframe batch from its per-view colour gradients (no counterpart in the reference, which trains
 * on one view per step):   dL_dsh[i][k][ch] = scale * sum_v basis_k(normalize(means3D[i] - campos[v])) * views[v][i][ch]
 * for k < (D+1)^2 and 0 for (D+1)^2 <= k < M; basis_k as computeColorFromSH (forward.cu:20-71).  views is
 * [n_views,P,3] (the dL_dcolor_view outputs of gsr_backward, gathered), campos [n_views,3] on the device; view_stride /
 * campos_stride = floats between consecutive views / centres (0 = dense: 3 P and 3), so that both may live in ONE gathered
 * buffer of [n_views, P + 1, 3] whose last row per view is the camera centre (a single all-gather); scale is 1/n_views
 * for the batch mean.  With one process per GPU this replaces the all-reduce of the [P,M,3] gradient
 * (2 x 192 B sent per Gaussian on a ring) by an all-gather of 12 B per Gaussian and view. */
int gsr_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                           long long campos_stride, const float* dL_dcolor_views, long long view_stride, float scale,
                           float* dL_dsh, void* stream);

/* The same with the optimizer fused in (as gsr_backward_args.sh_adam): instead of writing dL_dsh, this step's Adam update with
 * that batch-mean gradient is applied to shs [P,16,3] IN PLACE and to the two moment tensors (sh_adam->param must be shs or
 * NULL).  16-byte aligned [P,16,3] tensors only (GSR_ERR_UNSUPPORTED otherwise).  Reads means3D: call it before the
 * positions' own update. */
int gsr_sh_adam_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                           long long campos_stride, const float* dL_dcolor_views, long long view_stride, float scale,
                           float* shs, const gsr_sh_adam* sh_adam, void* stream);
/* With sh_adam->lazy set (gsr_sh_adam_lazy) the data-parallel step is as lean as the single-GPU one: a row whose colour
 * gradient is zero in EVERY gathered view would take a zero-gradient step -- it is left alone and steps later; a row some view
 * lights first takes the zero-gradient steps it is behind (this rank's forward pass only caught up the rows ITS view sees),
 * then this step, and row_step[i] = step.  1152 B of optimizer traffic per Gaussian some view of the batch sees instead of
 * per Gaussian.  The call may cover a row range (pointers offset by the caller, row_step too).  The lag of the rows nobody
 * lights is bounded either by gsr_backward (sh_adam together with dL_dcolor_view: the catch-up runs ahead of the exchange,
 * next to the blend kernel -- what both hosts do) or by gsr_sh_adam_lazy_slice over ALL rows after the last range of the
 * step (this step's 1/window of the row blocks catches up to `step`).  Same contract as the single-GPU lazy mode otherwise (pass the struct to
 * gsr_forward_args.sh_adam of the step; gsr_sh_adam_flush before anything else touches the tensor); results bit-identical to
 * the eager update (tests/test_lazy_sh_adam.py). */
/* PACKED exchange of the colour gradients.  Most rows of a view's [P,3] colour gradient are all-zero words (a view sees about
 * half of a 2 M-Gaussian map; the culled rest is zero), so a rank may send only the rows its view SEES (lit, or carrying the
 * -0.0f visibility marker) -- the same information in (1 bit + 1/16 word) per Gaussian + 12 B per seen Gaussian:
 *
 *   message (uint32 words; gsr_packed_view_words(P, capacity)):
 *     [0] K = seen rows   [1] P   [2] capacity (rows)   [3] 1 if K > capacity (rows beyond it were DROPPED: a caller's bug)
 *     [4..6] the view's camera centre (3 floats)   [7] 0
 *     prefix[ceil(P/64)]   seen rows in front of the 64-row group (exclusive)
 *     mask  [2 ceil(P/64)] bit (i % 64) of the 64-bit word i / 64: row i is seen (low word first)
 *     rows  [3 capacity]   the seen rows in index order (floats)
 *
 * gsr_pack_color_view builds one message from a view's dL_dcolor_view (four launches; scratch = gsr_pack_scratch_bytes(P));
 * `capacity` must be the same on every rank (the messages travel through ONE all-gather of equal chunks): the ranks agree on
 * max_v K_v beforehand -- K of a view is its number of visible Gaussians, which gsr_forward leaves for the calling thread in
 * gsr_last_visible_count().  gsr_sh_grad_from_packed_views / gsr_sh_adam_from_packed_views are gsr_sh_grad_from_views /
 * gsr_sh_adam_from_views on n_views such messages, msg_stride words apart: bit-identical results (the same rows, the same
 * order of the views).  At 2 M Gaussians, 47 % seen: 11.7 MB instead of 24 MB per rank on every link.
 * gsr_pack_view_plan writes the sections that do not depend on the gradient -- masks and prefix, from the radii of the forward
 * pass (seen = radii > 0; three launches, any stream once gsr_forward has returned) -- so that gsr_backward can write rows and
 * header itself (gsr_backward_args.packed_view). */
size_t gsr_packed_view_words(int P, int capacity_rows);
int gsr_pack_view_plan(int P, const int* radii, uint32_t* message, void* scratch, void* stream);
size_t gsr_pack_scratch_bytes(int P);
int gsr_pack_color_view(int P, const float* dL_dcolor_view, const float* campos, int capacity_rows, uint32_t* message, void* scratch,
                        void* stream);
int gsr_sh_grad_from_packed_views(int P, int D, int M, int n_views, const float* means3D, const uint32_t* messages,
                                  long long msg_stride, float scale, float* dL_dsh, void* stream);
int gsr_sh_adam_from_packed_views(int P, int D, int M, int n_views, const float* means3D, const uint32_t* messages,
                                  long long msg_stride, float scale, float* shs, const gsr_sh_adam* sh_adam, void* stream);
/* Gaussians with radii > 0 in the last gsr_forward of the calling thread (-1 before the first; 0 after a call with P == 0) */
int gsr_last_visible_count(void);
/* Diagnostic: forward passes of the calling thread whose depth sort ran a second time.  The depth sort takes three passes over 27
 * bits of (key - bits(0.2f)) -- every visible Gaussian has z > 0.2 -- and the host learns the largest key of the view with the
 * instance count; a view with a Gaussian at z >= 13 107 (or a depth that is not a number) is sorted again on all 32 bits.  The
 * results are the same either way; the second path costs about 0.15 ms at 2 M Gaussians. */
long long gsr_depth_resort_count(void);
/* Introspection: the binning arrangement gsr_forward takes for a view of this size with these raw_params bits (1 = tile-first,
 * 0 = depth-first; GSR_BINNING_* above, the GSR_BINNING environment override included). */
int gsr_binning_tile_first(int raw_params, int P, int width, int height);
/* The loud form of the decoders' silent guards (they decode nothing from a message whose P differs and read no row beyond the
 * rows a message holds): copies the n_views headers to the host, WAITS for `stream`, and returns GSR_ERR_INVALID_ARG unless
 * every message says P rows total, K <= min(its own capacity, `capacity_rows` = the rows that travelled) and "nothing dropped".  A synchronisation: for
 * tests, the first steps of a session and a debugging run -- not for every step. */
int gsr_check_packed_views(int P, int n_views, const uint32_t* messages, long long msg_stride, int capacity_rows, void* stream);
/* Diagnostic: the time the calling thread has spent BLOCKED in gsr_forward's one host synchronisation (the read of the instance
 * count behind the projection kernel) and the number of such waits, since the last reset.  A host that runs ahead of the device
 * waits there for most of a step; a wait near zero means the device is waiting for the HOST (launch-bound step: the gaps
 * between kernels are then host time, and every host-side call of the step costs step time). */
int gsr_host_wait_stats(double* total_us, long long* calls, int reset);

/* ahead == 0: after the last gsr_sh_adam_from_views range of the step -- row blocks b with b % window == step % window, every
 * row that is behind catches up to `step`.  ahead != 0 (window >= 3): BEFORE the step's gsr_sh_adam_from_views calls (what
 * gsr_backward does on its second stream in the view-factored mode) -- row blocks b with b % (window - 1) == step % (window - 1)
 * catch up to step - 1.  A step uses one of the two. */
int gsr_sh_adam_lazy_slice(int P, const gsr_sh_adam* adam, int ahead, void* stream);

/* Rasterizer::markVisible, cuda_rasterizer/rasterizer_impl.cu:141-153:
 * present[i] = (view-space z of means3D[i] > 0.2).  present is [P] bytes (bool). */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/* SimpleKNN::knn, third_party/simple-knn/simple_knn.cu:185-221: meanDists[i] = mean of
 * the squared distances from points[i] to its 3 nearest other points.  The reference
 * cudaMalloc's internally; here scratch comes from the callback like everything else. */
int gsr_knn_mean_dist2(int P, const float* points, float* meanDists,
                       gsr_alloc_fn scratchBuffer, void* scratch_ctx, void* stream);

/* ---- train-step kernels next to the rasterizer (the largest non-raster costs) ----
 *
 * Masked L1 + SSIM loss of GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:692-698,
 * include/loss_utils.h:28-124):  x = rendered * mask;
 *   loss = (1-lambda) * mean|x - gt| + lambda * (1 - mean(ssim_map(x, gt)))    (11x11 window, sigma 1.5)
 * Writes the scalar loss to *loss (device) and dloss/drendered to grad_rendered [3,H,W]; replaces
 * 5 + 10 grouped conv2d of the autograd graph.  mask may be NULL (all ones).
 * scratch: gsr_loss_scratch_bytes(width, height) device bytes. */
size_t gsr_loss_scratch_bytes(int width, int height);
int gsr_l1_ssim_loss(const float* rendered, const float* gt, const float* mask, int width, int height,
                     float lambda_dssim, float* grad_rendered, float* loss, char* scratch, void* stream);

/* One torch::optim::Adam step (no amsgrad / weight decay; src/gaussian_model.cpp:477-510 uses eps 1e-15)
 * on a flat fp32 tensor: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= (lr / (1-b1^step)) * m / (sqrt(v) / sqrt(1-b2^step) + eps).  The hyper-parameters are double as in
 * torch::optim::AdamOptions; the scalars derived from them (step size, 1-beta, 1/sqrt(bias correction 2)) are formed in
 * double and rounded once, the element arithmetic is fp32 (pinned to the reference's own trainingSetup +
 * torch::optim::Adam::step, tests/test_densify_reference.py).
 * period/split/lr_tail: if period > 0, elements [split, period) of every period-element row use lr_tail
 * (features_dc and features_rest live in one [P,16,3] buffer with learning rates lr and lr/20). */
int gsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                  double beta1, double beta2, double eps, int step, int period, int split, double lr_tail, void* stream);

/* The same step for several tensors in ONE launch (the four small per-Gaussian tensors of a data-parallel step, whose
 * gradients arrive together from one all-reduce): element arithmetic exactly as gsr_adam_step, one learning rate and step
 * counter per tensor.  count <= 8; tensors with n == 0 are skipped. */
typedef struct gsr_adam_multi_tensor {
	float* param;            /* UPDATED IN PLACE */
	const float* grad;
	float* exp_avg;
	float* exp_avg_sq;
	long long n;             /* elements */
	double lr;
	int step;                /* >= 1: the step being taken */
	float grad_scale;        /* the gradient is multiplied by this as it is read (1 = as it is): the 1/N of a batch mean whose
	                            all-reduce SUMMED -- one pass over the gradients less than averaging first */
} gsr_adam_multi_tensor;
int gsr_adam_step_multi(int count, const gsr_adam_multi_tensor* tensors, double beta1, double beta2, double eps, void* stream);

/* Per-view densification statistics (src/gaussian_mapper.cpp:714-719, src/gaussian_model.cpp:817-831) for
 * vis = radii > 0:  max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |dL_dmean2D.xy|; denom += 1.
 * dL_dmean2D is the [P,3] viewspace gradient; accum/denom are [P] ([P,1]) floats, max_radii2D [P] floats. */
int gsr_densify_stats(int P, const float* dL_dmean2D, const int* radii, float* xyz_gradient_accum, float* denom,
                      float* max_radii2D, void* stream);

/* ---- densification / pruning as stream compaction ----
 * GaussianModel::densifyAndPrune = densifyAndClone + densifyAndSplit (N = 2) + the final prunePoints
 * (src/gaussian_model.cpp:716-815) rebuild every tensor and Adam moment 4-6 times through boolean-mask indexing and cat.
 * Here: gsr_densify_select turns the per-Gaussian decisions into a gather plan with deterministic ballot/prefix ranks (no
 * atomics, no host round trip inside), the caller reads the six counts ONCE to size the new tensors, and
 * gsr_densify_gather rebuilds all five parameter tensors, their ten moment tensors and the statistics in one launch.
 * Order of the new set, as the reference's: [originals neither split nor pruned | surviving clones | surviving first
 * children | surviving second children], each block in source order.
 *
 * Decisions (all fp32, thresholds formed as the reference forms them):
 *   g = accum / denom, nan -> 0;  smax = max(exp(scaling));  big = smax > percent_dense * extent
 *   clone: sqrt(g*g) >= max_grad && !big          split: g >= max_grad && big
 *   pruned row: sigmoid(opacity) < min_opacity || (max_screen_size != 0 && its activated scale > 0.1f * extent)
 *   (clones inherit the source's decision; both children carry scale / 1.6; max_radii2D is zero at that point -- it was
 *   reset by densificationPostfix -- so the reference's screen-size term never fires)
 * prune_mask != NULL selects GaussianModel::prunePoints(mask) (:588-642) instead: keep = !mask, no clones, no children. */
typedef struct gsr_densify_select_args {
	int P;
	const float* xyz_gradient_accum;  /* [P] ([P,1]) */
	const float* denom;               /* [P] */
	const float* scaling;             /* [P,3] log-scales (the raw parameter) */
	const float* opacity;             /* [P] logits (the raw parameter) */
	float percent_dense, max_grad, min_opacity, extent;
	int max_screen_size;
	const uint8_t* prune_mask;        /* [P] bytes (bool) or NULL */
} gsr_densify_select_args;
size_t gsr_densify_scratch_bytes(int P);
/* counts: DEVICE int[8], written on the stream: [0] kept originals, [1] surviving clones, [2] surviving parents of
 * children (2 rows each), [3] split-selected parents k (the reference draws 2k x 3 normal samples, parent-major per copy),
 * [4] clone-selected, [5] rows of the new set = [0] + [1] + 2 [2].  scratch: gsr_densify_scratch_bytes(P) device bytes; it
 * holds the plan and must reach gsr_densify_gather unchanged. */
int gsr_densify_select(const gsr_densify_select_args* args, char* scratch, int* counts, void* stream);

typedef struct gsr_densify_gather_args {
	int P;                            /* rows of the source arrays (as passed to gsr_densify_select) */
	int n_new, n_keep, n_clone, n_child, n_split;   /* counts[5], [0], [1], [2], [3] read back by the caller */
	int features_row_floats;          /* 3 * M of the [P,M,3] SH tensor */
	/* index 0..4 = xyz [.,3], features [.,M,3], opacity [.,1], scaling [.,3], rotation [.,4]; moments may be NULL (in and
	 * out together) when no optimizer state exists.  In and out must not overlap. */
	const float* param_in[5];
	const float* exp_avg_in[5];
	const float* exp_avg_sq_in[5];
	float* param_out[5];
	float* exp_avg_out[5];
	float* exp_avg_sq_out[5];
	/* [2 n_split, 3] STANDARD normal draws: a child's offset is R(q_parent) * (z * exp(scaling_parent)), i.e.
	 * at::normal(0, std) of the reference (:731-736) is randn * std.  NULL allowed when n_child == 0. */
	const float* samples;
	float* stats_out[3];              /* xyz_gradient_accum, denom, max_radii2D of the new set ([n_new] each): zero-filled; NULL = skip */
	/* GaussianModel::exist_since_iter_ ([P] / [n_new] int32; both or neither): every row of the new set inherits its source's
	 * value -- prunePoints (:636), clones (:782) and split children (:744) alike */
	const int* exist_since_iter_in;
	int* exist_since_iter_out;
	/* Extension (NULL = the reference's row order: kept originals, clones, first children, second children, each in source order):
	 * gsr_densify_morton_scratch_bytes(n_new) device bytes -- the rows of the new set are then laid out along a Z-order curve of
	 * their (parents') positions.  The SAME Gaussians with the same values, moments, statistics and exist_since_iter, in another
	 * order (a view's depth ties then resolve by the new ids).  Why: every per-Gaussian kernel of a step fetches the rows of the
	 * Gaussians a view sees, a spatially compact subset; when neighbours in space are neighbours in memory those rows share 128-byte
	 * lines instead of being a random half of every line (measured on the synthetic cloud, whose ids carry no locality at all:
	 * -5 % of the train step; a SLAM map that grows keyframe by keyframe has some of it by construction).  The gather rewrites every
	 * tensor anyway: the order is free. */
	char* morton_scratch;
} gsr_densify_gather_args;
size_t gsr_densify_morton_scratch_bytes(int n_new);
int gsr_densify_gather(const gsr_densify_gather_args* args, const char* scratch, void* stream);

/* ---- Photo-SLAM's point-cloud kernels ----
 * transformPoints (src/operate_points.cu:73-93): out = M[:3,:4] * (p, 1), M = transformmatrix[16] with
 * element (r,c) at [4c+r] (transformPoint4x3). */
int gsr_transform_points(int P, const float* points, const float* transformmatrix, float* out_points, void* stream);
/* scale_and_transform_points (src/operate_points.cu:52-71): for mask[i] != 0: out_points[i] = M * (scale * p_i),
 * out_rots[i] = quaternion of (M[:3,:3] * R(q_i)), q stored (w, x, y, z).  Unmasked rows are NOT written.
 * reference_rot_layout != 0 reproduces insert_rot_to_rots exactly as shipped (cuda_rasterizer/operate_points.h:
 * 175-178 writes component +2 twice and never +3: the row becomes (w, x, z, <previous content>)); 0 writes the
 * intended (w, x, y, z). */
int gsr_scale_transform_points(int P, float scale, const float* points, const float* rots, const float* transformmatrix,
                               const uint8_t* mask, float* out_points, float* out_rots, int reference_rot_layout,
                               void* stream);
/* reproject_depths_pinhole (src/stereo_vision.cu:39-61): pixel i = (u = i % width, v = i / width), for mask[i]:
 * out = ((u-cx)*d/fx, (v-cy)*d/fy, d).  Unmasked rows are NOT written. */
int gsr_reproject_depth_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                const uint8_t* mask, float* out_points, void* stream);
/* search_neighborhood_to_estimate_depth_and_reproject_pinhole (src/stereo_vision.cu:63-136): keypoints with a 3D
 * point are copied; the others take the depth of the nearest (squared pixel distance <= max_pixel_dist, first wins)
 * keypoint that has one and are re-projected, or get z = -1.  colors is indexed exactly as the reference does
 * (colors[int(v*width+u) + 0..2]). */
int gsr_neighborhood_depth_pinhole(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                   const float* pixels, const uint8_t* has3D, const float* point3D, const float* colors,
                                   float* out_points, float* out_colors, void* stream);

/* Scratch sizes (bytes) gsr_forward will request, for callers that pre-allocate. */
size_t gsr_geometry_bytes(int P);
size_t gsr_binning_bytes(int num_rendered);
size_t gsr_image_bytes(int width, int height);
size_t gsr_knn_scratch_bytes(int P);

/* Optional per-stage timing with HIP events recorded on the caller's stream (process-wide switch,
 * meant for single-stream benchmarking).
 * After gsr_profile_enable(1), every gsr_forward / gsr_backward records events between its
 * stages; gsr_profile_read() waits for the last ones and returns milliseconds per stage
 * (-1 for stages that did not run or were not timed), indexed 0..gsr_profile_stage_count()-1.
 * gsr_profile_enable(2) times only the backward blend (two event records per step instead of
 * eleven: every record is a ~5 us bubble in the stream); gsr_profile_enable(0) switches off. */
int gsr_profile_enable(int on);
int gsr_profile_stage_count(void);
const char* gsr_profile_stage_name(int stage);
int gsr_profile_read(float* ms, int count);

const char* gsr_strerror(int status);
/* hipError_t of the last failing HIP call on this thread (0 if none), and its name. */
int gsr_last_hip_error(void);
const char* gsr_last_hip_error_string(void);
/* "hip-gfx950" for the product library. */
const char* gsr_backend(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
