/*
 * gsr_oracle.h -- CPU restatement ("oracle") of the Photo-SLAM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the CPU baseline -- never as the thing shipped or measured
 * as the GPU path.
 *
 * PINNING.  The reference (HuajianUP/Photo-SLAM) ships no tests, golden vectors or
 * fixtures for this path, and it is CUDA-only.  Every function of this oracle is
 * pinned against the reference's OWN sources compiled for the host with g++ against
 * small shims of the CUDA block model, the CUB calls, thrust::device_vector and the
 * glm operators they use (oracle/build_ref.py, oracle/ref_shim/ -> oracle/_ref/):
 *   cuda_rasterizer/{forward,backward,rasterizer_impl}.cu   gsro_forward / _backward / _mark_visible
 *   third_party/simple-knn/simple_knn.cu                     gsro_knn
 *   the __global__ kernels of src/operate_points.cu, src/stereo_vision.cu (+ cuda_rasterizer/
 *   operate_points.h, stereo_vision.h)                       the four point kernels
 * tests/test_reference_pinning.py requires every forward quantity -- radii, tiles, sort
 * keys, instance list, ranges, n_contrib, final T, image, kNN distances, points -- to
 * agree BIT FOR BIT and the gradients to 1e-5, on random scenes and on the committed
 * fixture tests/golden/reference_small.npz (which also checks the HIP kernels directly,
 * on the GPU boxes where the reference tree does not exist).  What no host build
 * reproduces: nvcc's FMA contraction and the GPU's float-atomic order; the shims restate
 * the published semantics of CUB (stable LSD sort, inclusive sum, reduce) and of glm
 * 0.9.9 (column-major mat3 product, dot, length), which the reference does not vendor.
 * Further independent checks: analytic known-answer tests, a float64 autograd renderer
 * and brute-force kNN (tests/test_oracle_pinning.py).
 */
#ifndef GSR_ORACLE_H
#define GSR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Every intermediate the reference keeps in GeometryState / BinningState /
 * ImageState (cuda_rasterizer/rasterizer_impl.h:32-62), exposed for stage tests. */
typedef struct gsro_state {
	int P, D, M, W, H, grid_x, grid_y, R, sort_bits;
	/* GeometryState */
	float*    depths;        /* [P]   */
	uint8_t*  clamped;       /* [3P]  */
	int*      radii;         /* [P]   */
	float*    means2D;       /* [2P]  */
	float*    cov3D;         /* [6P]  */
	float*    conic_opacity; /* [4P]  */
	float*    rgb;           /* [3P]  */
	uint32_t* tiles_touched; /* [P]   */
	uint32_t* point_offsets; /* [P] inclusive scan */
	/* BinningState */
	uint64_t* keys_unsorted; /* [R] */
	uint32_t* vals_unsorted; /* [R] */
	uint64_t* keys_sorted;   /* [R] */
	uint32_t* point_list;    /* [R] */
	/* ImageState */
	uint32_t* ranges;        /* [2T] (x=start,y=end) */
	float*    final_T;       /* [W*H] */
	uint32_t* n_contrib;     /* [W*H] */
	/* oracle-only: pixels where a skip/terminate decision sits within rounding
	 * noise of its threshold (exp() ulp, fma contraction); excluded from the
	 * bit-exact n_contrib comparison. */
	uint8_t*  fragile;       /* [W*H] */
} gsro_state;

/* threads <= 0 -> all cores (OpenMP); 1 -> serial, fully deterministic. */
void gsro_set_threads(int threads);
int  gsro_get_threads(void);

/* Rasterizer::forward, cuda_rasterizer/rasterizer_impl.cu:198-336.
 * nullptr for an absent optional (shs / colors_precomp / scales+rotations /
 * cov3D_precomp), exactly as the reference.  out_color[3*H*W] and radii[P] are
 * written.  Returns a state to be released with gsro_free(), or NULL. */
gsro_state* gsro_forward(
	int P, int D, int M,
	const float* background, int W, int H,
	const float* means3D, const float* shs, const float* colors_precomp,
	const float* opacities, const float* scales, float scale_modifier,
	const float* rotations, const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos,
	float tan_fovx, float tan_fovy, int prefiltered,
	float* out_color, int* radii);

/* Rasterizer::backward, cuda_rasterizer/rasterizer_impl.cu:340-433.  All
 * gradient arrays must be zero-initialised by the caller (the reference uses
 * torch::zeros, src/rasterize_points.cu:149-157).  dL_dmean2D is [P,3],
 * dL_dconic is [P,4] (the [P,2,2] tensor; .z never written).
 * Gradient sums over pixels are accumulated in double and rounded once:
 * the reference's float atomicAdd order is undefined (backward.cu:523-554), so
 * the oracle reports the order-free value every valid fp32 order rounds about. */
void gsro_backward(
	const gsro_state* st,
	const float* background,
	const float* means3D, const float* shs, const float* colors_precomp,
	const float* scales, float scale_modifier, const float* rotations,
	const float* cov3D_precomp,
	const float* viewmatrix, const float* projmatrix, const float* cam_pos,
	float tan_fovx, float tan_fovy,
	const float* dL_dpix,
	float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
	float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
	float* dL_drot);

void gsro_free(gsro_state* st);

/* Rasterizer::markVisible, rasterizer_impl.cu:141-153 (+ checkFrustum :54-66). */
void gsro_mark_visible(int P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, uint8_t* present);

/* SimpleKNN::knn, third_party/simple-knn/simple_knn.cu:185-221: the Morton/box
 * algorithm restated (meanDists[P]). */
void gsro_knn(int P, const float* points, float* meanDists);
/* Brute-force O(P^2) exact 3-NN mean of squared distances (pins gsro_knn). */
void gsro_knn_bruteforce(int P, const float* points, float* meanDists);

/* Photo-SLAM point kernels: src/operate_points.cu:38-71 (+ cuda_rasterizer/operate_points.h:39-179) and
 * src/stereo_vision.cu:39-136 (+ cuda_rasterizer/stereo_vision.h:39-53).  Outputs must be pre-zeroed by the caller
 * like the reference's torch::zeros_like; unmasked rows are left untouched. */
void gsro_transform_points(int P, const float* pts, const float* m, float* out);
void gsro_scale_transform_points(int P, float scale, const float* pts, const float* rots, const float* m,
                                 const uint8_t* mask, float* out_pts, float* out_rots, int reference_rot_layout);
void gsro_reproject_depth_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                  const uint8_t* mask, float* points);
void gsro_neighborhood_depth_pinhole(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                     const float* pixels, const uint8_t* has3D, const float* p3d, const float* colors,
                                     float* out_p, float* out_c);

/* Analysis helper (not in the reference): work statistics of the per-quad rejection of the HIP
 * blend kernels, and the count of blended pairs it would wrongly reject (must be 0).  out[16]. */
void gsro_cull_stats(const gsro_state* st, double* out);

/* getHigherMsb, rasterizer_impl.cu:35-50. */
uint32_t gsro_higher_msb(uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
