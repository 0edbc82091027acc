/*
 * ref_api.cpp -- C entry points around the REFERENCE's CudaRasterizer::Rasterizer (cuda_rasterizer/rasterizer.h:24-82),
 * built by oracle/build_ref.py from the reference's own sources against the host shims in oracle/ref_shim/.
 * TEST INFRASTRUCTURE ONLY: pins the CPU oracle and the committed golden fixtures to the reference's code.
 */
#include <stdint.h>
#include <string.h>

#include <functional>
#include <vector>

#include "rasterizer_impl.h"   /* the reference's own header: state structs + Rasterizer */
#include "simple_knn.h"        /* third_party/simple-knn: SimpleKNN::knn */

namespace {
struct Buffers {
	std::vector<char> geom, binning, img;
};
std::function<char*(size_t)> resizer(std::vector<char>& v)
{
	return [&v](size_t n) {
		v.assign(n + 256, 0);
		return v.data();
	};
}
}  // namespace

/* Photo-SLAM point kernels, launched as their LibTorch wrappers do (src/operate_points.cu:85,128, src/stereo_vision.cu:159,201);
 * outputs are zero-initialised by the caller like the wrappers' torch::zeros_like. */
__global__ void transform_points(int P, const float* orig_points, const float* transformmatrix, float* trans_points);
__global__ void scale_and_transform_points(int P, const float scale, const float* orig_points, const float* orig_rots,
                                           const float* transformmatrix, const bool* mask, float* trans_points, float* trans_rots);
__global__ void reproject_depths_pinhole(int P, const int width, const float fx, const float fy, const float cx, const float cy,
                                         const float* depths, const bool* mask, float* points);
__global__ void search_neighborhood_to_estimate_depth_and_reproject_pinhole(int N, int width, const float fx, const float fy,
                                                                            const float cx, const float cy,
                                                                            const float max_pixel_dist, const float* pixels,
                                                                            const bool* has3D, const float* point3D_orig,
                                                                            const float* colors, float* point3D_result,
                                                                            float* colors_result);

extern "C" {

struct ref_state {
	int P, W, H, R;
	Buffers* buf;
	/* views into the reference's own state structs (rasterizer_impl.h:32-62) */
	float* depths; bool* clamped; int* radii; float* means2D; float* cov3D; float* conic_opacity; float* rgb;
	uint32_t* point_offsets; uint32_t* tiles_touched;
	uint64_t* keys_unsorted; uint64_t* keys_sorted; uint32_t* vals_unsorted; uint32_t* point_list;
	uint32_t* ranges; uint32_t* n_contrib; float* accum_alpha;
};

ref_state* ref_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii)
{
	ref_state* st = new ref_state();
	memset(st, 0, sizeof(*st));
	st->buf = new Buffers();
	st->P = P; st->W = W; st->H = H;
	st->R = CudaRasterizer::Rasterizer::forward(resizer(st->buf->geom), resizer(st->buf->binning), resizer(st->buf->img), P, D, M,
	                                            background, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
	                                            rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
	                                            prefiltered != 0, out_color, radii);
	char* c = st->buf->geom.data();
	CudaRasterizer::GeometryState g = CudaRasterizer::GeometryState::fromChunk(c, (size_t)P);
	st->depths = g.depths; st->clamped = g.clamped; st->radii = g.internal_radii; st->means2D = (float*)g.means2D;
	st->cov3D = g.cov3D; st->conic_opacity = (float*)g.conic_opacity; st->rgb = g.rgb; st->point_offsets = g.point_offsets;
	st->tiles_touched = g.tiles_touched;
	c = st->buf->img.data();
	CudaRasterizer::ImageState im = CudaRasterizer::ImageState::fromChunk(c, (size_t)W * H);
	st->ranges = (uint32_t*)im.ranges; st->n_contrib = im.n_contrib; st->accum_alpha = im.accum_alpha;
	if (st->R > 0) {
		c = st->buf->binning.data();
		CudaRasterizer::BinningState b = CudaRasterizer::BinningState::fromChunk(c, (size_t)st->R);
		st->keys_unsorted = b.point_list_keys_unsorted; st->keys_sorted = b.point_list_keys;
		st->vals_unsorted = b.point_list_unsorted; st->point_list = b.point_list;
	}
	return st;
}

void ref_backward(ref_state* st, int D, int M, const float* background, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                  float tan_fovx, float tan_fovy, const int* radii, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                  float* dL_drot)
{
	CudaRasterizer::Rasterizer::backward(st->P, D, M, st->R, background, st->W, st->H, means3D, shs, colors_precomp, scales,
	                                     scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx,
	                                     tan_fovy, radii, st->buf->geom.data(), st->buf->binning.data(), st->buf->img.data(),
	                                     dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
	                                     dL_dscale, dL_drot);
}

void ref_mark_visible(int P, float* means3D, float* viewmatrix, float* projmatrix, uint8_t* present)
{
	CudaRasterizer::Rasterizer::markVisible(P, means3D, viewmatrix, projmatrix, (bool*)present);
}

void ref_transform_points(int P, const float* pts, const float* m, float* out)
{
	if (P) CUDAEMU_LAUNCH((transform_points), (P + 255) / 256, 256, P, pts, m, out);
}
void ref_scale_transform_points(int P, float scale, const float* pts, const float* rots, const float* m, const uint8_t* mask,
                                float* out_pts, float* out_rots)
{
	if (P) CUDAEMU_LAUNCH((scale_and_transform_points), (P + 255) / 256, 256, P, scale, pts, rots, m, (const bool*)mask, out_pts, out_rots);
}
void ref_reproject_depth_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths, const uint8_t* mask,
                                 float* points)
{
	if (P) CUDAEMU_LAUNCH((reproject_depths_pinhole), (P + 255) / 256, 256, P, width, fx, fy, cx, cy, depths, (const bool*)mask, points);
}
void ref_neighborhood_depth_pinhole(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                    const float* pixels, const uint8_t* has3D, const float* p3d, const float* colors, float* out_p,
                                    float* out_c)
{
	if (N)
		CUDAEMU_LAUNCH((search_neighborhood_to_estimate_depth_and_reproject_pinhole), (N + 255) / 256, 256, N, width, fx, fy, cx, cy,
		               max_pixel_dist, pixels, (const bool*)has3D, p3d, colors, out_p, out_c);
}

/* SimpleKNN::knn, third_party/simple-knn/simple_knn.cu:185-221 */
void ref_knn(int P, float* points, float* meanDists) { SimpleKNN::knn(P, (float3*)points, meanDists); }

void ref_free(ref_state* st)
{
	if (!st) return;
	delete st->buf;
	delete st;
}

}  // extern "C"
