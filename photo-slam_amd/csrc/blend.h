// blend.h -- pieces shared by the forward and backward alpha-blend kernels.
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

// Workgroup = one 16x16 tile (the tile grid is part of the parity contract); wave q of the
// workgroup owns the 8x8 pixel quad q (quad origin = ((q&1)*8, (q>>1)*8)), lane l the pixel
// (l&7, l>>3) inside it.  8x8 quads are the most compact 64-pixel footprint, which is what
// makes the per-quad rejection below effective.
__device__ __forceinline__ void quad_pixel(int tile_x, int tile_y, int& px, int& py)
{
	const int q = wave_id(), l = lane_id();
	px = tile_x * TILE + (q & 1) * 8 + (l & 7);
	py = tile_y * TILE + (q >> 1) * 8 + (l >> 3);
}

// XCD-aware tile assignment: workgroup b is dispatched to XCD b % 8 (observed, used for
// speed only), so XCD k is given a contiguous band of tiles -- neighbouring tiles share
// Gaussian records, which then stay in that XCD's 4 MiB L2.
__device__ __forceinline__ int xcd_tile(int block, int tiles)
{
	const int per = (tiles + 7) >> 3;
	return (block & 7) * per + (block >> 3);
}
static inline int xcd_grid(int tiles) { return ((tiles + 7) >> 3) * 8; }

// Conservative per-quad rejection.  A (pixel, Gaussian) pair is skipped by the reference
// when alpha = min(0.99, o*exp(power)) < 1/255 (forward.cu:343-345, backward.cu:499-501),
// i.e. when q = -power > ln(255*o).  For a positive-definite conic q is convex, so its
// minimum over the quad's pixel rectangle is 0 if the centre lies inside and otherwise sits
// on one of the four edges (1-D clamped parabola minimum per edge).  If even that minimum
// exceeds the threshold (plus a rounding margin) no pixel of the quad can blend this
// Gaussian and the whole wave skips it -- the skipped pairs are exactly pairs the
// reference `continue`s over, so results are unchanged.
// Returns a 4-bit mask: bit q set = quad q must evaluate this Gaussian.
__device__ __forceinline__ uint32_t quad_keep_bits(const float4 q0, const float4 q1, float tile_px0, float tile_py0)
{
	const float mx = q0.x, my = q0.y, A = q0.z, B = q0.w, C = q1.x, o = q1.y;
	if (o < 1.0f / 255.0f) return 0u;             // alpha <= o < 1/255 everywhere (NaN opacity falls through: keep)
	const float det = A * C - B * B;
	if (!(A > 0.f && C > 0.f && det > 0.f)) return 0xFu;  // not positive definite (or NaN): no bound, keep
	const float thr = __logf(255.0f * o);
	uint32_t bits = 0;
	const float invA = 1.0f / A, invC = 1.0f / C;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const float u0 = tile_px0 + (float)((q & 1) * 8) - mx, u1 = u0 + 7.0f;
		const float v0 = tile_py0 + (float)((q >> 1) * 8) - my, v1 = v0 + 7.0f;
		float qmin;
		if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) {
			qmin = 0.f;
		} else {
			// edge u = const: minimise over v in [v0, v1];  edge v = const: minimise over u
			float e;
			float vs = fminf(v1, fmaxf(v0, -B * u0 * invC));
			qmin = 0.5f * (A * u0 * u0 + C * vs * vs) + B * u0 * vs;
			vs = fminf(v1, fmaxf(v0, -B * u1 * invC));
			e = 0.5f * (A * u1 * u1 + C * vs * vs) + B * u1 * vs;
			qmin = fminf(qmin, e);
			float us = fminf(u1, fmaxf(u0, -B * v0 * invA));
			e = 0.5f * (A * us * us + C * v0 * v0) + B * us * v0;
			qmin = fminf(qmin, e);
			us = fminf(u1, fmaxf(u0, -B * v1 * invA));
			e = 0.5f * (A * us * us + C * v1 * v1) + B * us * v1;
			qmin = fminf(qmin, e);
		}
		// rounding margin: fp32 evaluation error of `power` scales with the magnitude of its terms
		const float um = fmaxf(fabsf(u0), fabsf(u1)), vm = fmaxf(fabsf(v0), fabsf(v1));
		const float mag = 0.5f * (A * um * um + C * vm * vm) + fabsf(B) * um * vm;
		const float margin = 0.01f + 1e-4f * thr + 2e-5f * mag;
		if (!(qmin > thr + margin)) bits |= (1u << q);
	}
	return bits;
}

}  // namespace gsr
