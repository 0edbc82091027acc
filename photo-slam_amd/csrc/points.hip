// points.hip -- Photo-SLAM's point-cloud kernels next to the rasterizer:
//   transform_points / scale_and_transform_points         src/operate_points.cu:38-71
//   (helpers transform_point, scale_and_transform_point, transfrom_quaternion_using_matrix,
//    insert_rot_to_rots                                    cuda_rasterizer/operate_points.h:39-179)
//   reproject_depths_pinhole / search_neighborhood_...     src/stereo_vision.cu:39-136
// Small streaming kernels (12-28 B per point); compiled -ffp-contract=off so results are bit-identical
// to the CPU oracle.  The reference's quirks are kept where a caller could observe them and are
// switchable where they are plain bugs (see gsr.h).
#include "state.h"
#include "wave64.h"

#include <float.h>

namespace gsr {

__device__ __forceinline__ void xform4x3(const float* m, float x, float y, float z, float& ox, float& oy, float& oz)
{
	// transformPoint4x3, auxiliary.h:58-66
	ox = m[0] * x + m[4] * y + m[8] * z + m[12];
	oy = m[1] * x + m[5] * y + m[9] * z + m[13];
	oz = m[2] * x + m[6] * y + m[10] * z + m[14];
}

__global__ void __launch_bounds__(256)
transform_points_kernel(int P, const float* __restrict__ pts, const float* __restrict__ m, float* __restrict__ out)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= P) return;
	float x, y, z;
	xform4x3(m, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], x, y, z);
	out[3 * (size_t)i] = x;
	out[3 * (size_t)i + 1] = y;
	out[3 * (size_t)i + 2] = z;
}

__global__ void __launch_bounds__(256)
scale_transform_points_kernel(int P, float scale, const float* __restrict__ pts, const float* __restrict__ rots,
                              const float* __restrict__ m, const uint8_t* __restrict__ mask, float* __restrict__ out_pts,
                              float* __restrict__ out_rots, int reference_rot_layout)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= P || !mask[i]) return;
	float x, y, z;
	xform4x3(m, pts[3 * (size_t)i] * scale, pts[3 * (size_t)i + 1] * scale, pts[3 * (size_t)i + 2] * scale, x, y, z);
	out_pts[3 * (size_t)i] = x;
	out_pts[3 * (size_t)i + 1] = y;
	out_pts[3 * (size_t)i + 2] = z;

	// transfrom_quaternion_using_matrix, operate_points.h:72-156: stored order is (w, x, y, z)
	const float qw = rots[4 * (size_t)i], qx = rots[4 * (size_t)i + 1], qy = rots[4 * (size_t)i + 2], qz = rots[4 * (size_t)i + 3];
	const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
	const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
	const float txx = tx * qx, txy = ty * qx, txz = tz * qx;
	const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
	const float R00 = 1.0f - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
	const float R10 = txy + twz, R11 = 1.0f - (txx + tzz), R12 = tyz - twx;
	const float R20 = txz - twy, R21 = tyz + twx, R22 = 1.0f - (txx + tyy);
	float R[3][3];
	R[0][0] = m[0] * R00 + m[4] * R10 + m[8] * R20;
	R[0][1] = m[0] * R01 + m[4] * R11 + m[8] * R21;
	R[0][2] = m[0] * R02 + m[4] * R12 + m[8] * R22;
	R[1][0] = m[1] * R00 + m[5] * R10 + m[9] * R20;
	R[1][1] = m[1] * R01 + m[5] * R11 + m[9] * R21;
	R[1][2] = m[1] * R02 + m[5] * R12 + m[9] * R22;
	R[2][0] = m[2] * R00 + m[6] * R10 + m[10] * R20;
	R[2][1] = m[2] * R01 + m[6] * R11 + m[10] * R21;
	R[2][2] = m[2] * R02 + m[6] * R12 + m[10] * R22;
	float ow, ox, oy, oz;
	float t = R[0][0] + R[1][1] + R[2][2];
	if (t > 0.0f) {  // Shoemake, "Quaternion Calculus and Fast Animation"
		t = sqrtf(t + 1.0f);
		ow = 0.5f * t;
		t = 0.5f / t;
		ox = (R[2][1] - R[1][2]) * t;
		oy = (R[0][2] - R[2][0]) * t;
		oz = (R[1][0] - R[0][1]) * t;
	} else {
		int a = 0;
		if (R[1][1] > R[0][0]) a = 1;
		if (R[2][2] > R[a][a]) a = 2;
		const int b = (a + 1) % 3, c = (b + 1) % 3;
		t = sqrtf(R[a][a] - R[b][b] - R[c][c] + 1.0f);
		float xyz[3];
		xyz[a] = 0.5f * t;
		t = 0.5f / t;
		ow = (R[c][b] - R[b][c]) * t;
		xyz[b] = (R[b][a] + R[a][b]) * t;
		xyz[c] = (R[c][a] + R[a][c]) * t;
		ox = xyz[0];
		oy = xyz[1];
		oz = xyz[2];
	}
	out_rots[4 * (size_t)i] = ow;
	out_rots[4 * (size_t)i + 1] = ox;
	if (reference_rot_layout) {
		// insert_rot_to_rots writes index +2 twice and never +3 (operate_points.h:175-178): (w, x, z, <untouched>)
		out_rots[4 * (size_t)i + 2] = oz;
	} else {
		out_rots[4 * (size_t)i + 2] = oy;
		out_rots[4 * (size_t)i + 3] = oz;
	}
}

// reproject_depths_pinhole, stereo_vision.cu:39-61 (+ reproject_depth_pinhole, stereo_vision.h:39-53)
__global__ void __launch_bounds__(256)
reproject_depth_kernel(int P, int width, float fx, float fy, float cx, float cy, const float* __restrict__ depths,
                       const uint8_t* __restrict__ mask, float* __restrict__ points)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= P || !mask[i]) return;
	const int v = i / width, u = i - v * width;
	const float d = depths[i];
	points[3 * (size_t)i] = (u - cx) * d / fx;
	points[3 * (size_t)i + 1] = (v - cy) * d / fy;
	points[3 * (size_t)i + 2] = d;
}

// search_neighborhood_to_estimate_depth_and_reproject_pinhole, stereo_vision.cu:63-136.  Thread = keypoint;
// candidates are staged through LDS in tiles of 256 and scanned in index order, so the first nearest
// candidate wins exactly as in the reference's serial loop.
__global__ void __launch_bounds__(256)
neighborhood_depth_kernel(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                          const float* __restrict__ pixels, const uint8_t* __restrict__ has3D,
                          const float* __restrict__ p3d, const float* __restrict__ colors, float* __restrict__ out_p,
                          float* __restrict__ out_c)
{
	__shared__ float s_u[256], s_v[256], s_z[256];
	__shared__ uint8_t s_has[256];
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const bool valid = idx < N;
	const float u = valid ? pixels[2 * (size_t)idx] : 0.f, v = valid ? pixels[2 * (size_t)idx + 1] : 0.f;
	const bool mine3d = valid && has3D[idx];
	float min_dist = FLT_MAX, depth = -1.0f;   // MAXFLOAT
	for (int base = 0; base < N; base += 256) {
		const int j = base + (int)threadIdx.x;
		__syncthreads();
		if (j < N) {
			s_u[threadIdx.x] = pixels[2 * (size_t)j];
			s_v[threadIdx.x] = pixels[2 * (size_t)j + 1];
			s_z[threadIdx.x] = p3d[3 * (size_t)j + 2];
			s_has[threadIdx.x] = has3D[j];
		}
		__syncthreads();
		if (valid && !mine3d) {
			const int n = min(256, N - base);
			for (int k = 0; k < n; k++) {
				if (!s_has[k] || base + k == idx) continue;
				const float du = u - s_u[k], dv = v - s_v[k];
				const float dist = du * du + dv * dv;
				if (dist > max_pixel_dist || dist >= min_dist) continue;
				min_dist = dist;
				depth = s_z[k];
			}
		}
	}
	if (!valid) return;
	const size_t pt = 3 * (size_t)idx;
	const int px_in_image = (int)(v * width + u);   // float arithmetic truncated, as the reference
	if (mine3d) {
		out_p[pt] = p3d[pt];
		out_p[pt + 1] = p3d[pt + 1];
		out_p[pt + 2] = p3d[pt + 2];
		out_c[pt] = colors[px_in_image];
		out_c[pt + 1] = colors[px_in_image + 1];
		out_c[pt + 2] = colors[px_in_image + 2];
		return;
	}
	if (depth > 0.0f) {
		const int ui = (int)u, vi = (int)v;   // reproject_depth_pinhole takes int u, v
		out_p[pt] = (ui - cx) * depth / fx;
		out_p[pt + 1] = (vi - cy) * depth / fy;
		out_p[pt + 2] = depth;
		out_c[pt] = colors[px_in_image];
		out_c[pt + 1] = colors[px_in_image + 1];
		out_c[pt + 2] = colors[px_in_image + 2];
	} else {
		out_p[pt + 2] = -1.0f;
	}
}

}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_transform_points(int P, const float* points, const float* transformmatrix, float* out_points, void* stream_)
{
	if (P < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!points || !transformmatrix || !out_points) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	GSR_LAUNCH(transform_points_kernel, div_up(P, 256), 256, stream, P, points, transformmatrix, out_points);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int gsr_scale_transform_points(int P, float scale, const float* points, const float* rots, const float* transformmatrix,
                               const uint8_t* mask, float* out_points, float* out_rots, int reference_rot_layout,
                               void* stream_)
{
	if (P < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!points || !rots || !transformmatrix || !mask || !out_points || !out_rots) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	GSR_LAUNCH(scale_transform_points_kernel, div_up(P, 256), 256, stream, P, scale, points, rots, transformmatrix, mask,
	           out_points, out_rots, reference_rot_layout);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int gsr_reproject_depth_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                const uint8_t* mask, float* out_points, void* stream_)
{
	if (P < 0 || width <= 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!depths || !mask || !out_points) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	GSR_LAUNCH(reproject_depth_kernel, div_up(P, 256), 256, stream, P, width, fx, fy, cx, cy, depths, mask, out_points);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int gsr_neighborhood_depth_pinhole(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                   const float* pixels, const uint8_t* has3D, const float* point3D, const float* colors,
                                   float* out_points, float* out_colors, void* stream_)
{
	if (N < 0 || width <= 0) return GSR_ERR_INVALID_ARG;
	if (N == 0) return GSR_OK;
	if (!pixels || !has3D || !point3D || !colors || !out_points || !out_colors) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	GSR_LAUNCH(neighborhood_depth_kernel, div_up(N, 256), 256, stream, N, width, fx, fy, cx, cy, max_pixel_dist, pixels,
	           has3D, point3D, colors, out_points, out_colors);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // extern "C"
