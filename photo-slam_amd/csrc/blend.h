// blend.h -- pieces shared by the forward and backward alpha-blend kernels.
//
// Work decomposition: the 16x16 tile stays the unit of binning (tile ids are part of the parity
// contract), but the unit of execution is one WAVE per 8x8 pixel quad: a 64-thread workgroup
// owns quad q of a tile, lane l the pixel (l&7, l>>3) inside it.  The four quads of a tile walk
// the same depth-sorted list independently -- no workgroup barriers.  Per batch of 64 list
// entries lane j gathers entry j's 48-byte record, evaluates the per-quad rejection test for it
// and parks the (pre-scaled) record in the wave's 3 KiB LDS slice; a ballot yields the 64-bit
// survivor mask in an SGPR pair; the wave walks the set bits with scalar instructions and
// fetches each survivor with LDS broadcast reads (same address in all lanes: conflict-free, and
// on the LDS pipe instead of the VALU the blend math saturates).
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

constexpr int QUADS_PER_TILE = 4;
constexpr float LOG2E = 1.4426950408889634f;

// exp(power) = exp2(pw) with pw = A'dx^2 + C'dy^2 + B'dxdy and the conic pre-scaled once per staged
// entry (A' = -0.5*log2(e)*A, C' likewise, B' = -log2(e)*B): the per-pixel chain is 8 VALU + v_exp_f32.
__device__ __forceinline__ float4 prescale_q0(const float4 q0) { return make_float4(q0.x, q0.y, -0.5f * LOG2E * q0.z, -LOG2E * q0.w); }
__device__ __forceinline__ float prescale_c(float c) { return -0.5f * LOG2E * c; }

// XCD-aware work assignment: workgroup b is dispatched to XCD b % 8 (observed behaviour, used for speed only), every XCD has its
// own 4 MiB L2, and the dispatcher deals workgroups IN ORDER -- the XCDs advance in lockstep through their shares and the kernel
// ends with the XCD that holds the most work.  The tiles are dealt to the XCDs in CHUNKS, round robin (chunk c -> XCD c % 8);
// inside an XCD the chunks follow one another and the four quads of a tile are consecutive workgroups.
//   mode > 0   chunks of `mode` consecutive tiles of the row-major order: eight samples of the whole image per round of the deal --
//              the default, 8 tiles
//   mode == 0  ONE chunk per XCD: a contiguous band of the image -- the arrangement until round 5: eight different REGIONS of the
//              image (at C3 the heaviest holds 4.6 % more list entries than the mean, at C2 24 %)
// Measured at C3 (gsr_api.hip: xcd_deal_mode): bands -> row-major chunks of 8: blend_fwd 0.191 -> 0.173 ms, blend_bwd 0.451 -> 0.430.
// (Squares of e x e tiles with the backward blend taking the heaviest squares first -- filed under work classes by the forward
// blend -- were built in round 5 and measured again in round 6 at C2 / C4 / C5: -3 ... -7 % of blend_bwd, +3 % of blend_fwd, twice
// the record re-reads; removed: EXPERIMENTS.md.)
// (struct TileDeal, make_tile_deal: state.h)
__device__ __forceinline__ int xcd_tile(int in_xcd, int xcd, const TileDeal& d)
{
	if (d.mode == 0) {
		const int per = (d.tiles + 7) >> 3;
		return in_xcd >= per ? d.tiles : xcd * per + in_xcd;
	}
	const int t = ((in_xcd / d.mode) * 8 + xcd) * d.mode + in_xcd % d.mode;
	return t < d.tiles ? t : d.tiles;
}
static inline int xcd_tiles_per_xcd(const TileDeal& d)
{
	if (d.mode == 0) return (d.tiles + 7) >> 3;
	return ((((d.tiles + d.mode - 1) / d.mode) + 7) >> 3) * d.mode;
}
__device__ __forceinline__ int tile_assignment(int block, const TileDeal& d) { return xcd_tile(block >> 3, block & 7, d); }
static inline int tile_grid(const TileDeal& d) { return xcd_tiles_per_xcd(d) * 8; }
__device__ __forceinline__ void quad_assignment(int block, const TileDeal& d, int& tile, int& quad)
{
	const int in_xcd = block >> 3;
	tile = xcd_tile(in_xcd >> 2, block & 7, d);   // d.tiles: a padding workgroup
	quad = in_xcd & 3;
}
static inline int quad_grid(const TileDeal& d) { return xcd_tiles_per_xcd(d) * 8 * QUADS_PER_TILE; }

// Conservative per-quad rejection.  A (pixel, Gaussian) pair is skipped by the reference
// when alpha = min(0.99, o*exp(power)) < 1/255 (forward.cu:343-345, backward.cu:499-501),
// i.e. when q = -power > ln(255*o).  For a positive-definite conic q is convex, so its
// minimum over the quad's pixel rectangle is 0 if the centre lies inside and otherwise sits
// on one of the four edges (1-D clamped parabola minimum per edge).  If even that minimum
// exceeds the threshold (plus a rounding margin) no pixel of the quad can blend this
// Gaussian and the whole wave skips it -- the skipped pairs are exactly pairs the
// reference `continue`s over, so results are unchanged.
// (x0, y0) = pixel coordinates of the rectangle's first pixel, ext = its width and height - 1 (7: a quad; 15: a whole tile --
// the instance emission's test under GSR_CULL_EMPTY_TILES: a tile's rectangle contains its four quads', and its margin is the
// larger one, so an instance it rejects is one quad_keep rejects in all four quads).
__device__ __forceinline__ bool rect_keep(const float4 q0, const float4 q1, float x0, float y0, float ext)
{
	const float mx = q0.x, my = q0.y, A = q0.z, B = q0.w, C = q1.x, o = q1.y;
	if (o < 1.0f / 255.0f) return false;          // alpha <= o < 1/255 everywhere (NaN opacity falls through: keep)
	const float det = A * C - B * B;
	if (!(A > 0.f && C > 0.f && det > 0.f)) return true;  // not positive definite (or NaN): no bound, keep
	const float thr = __logf(255.0f * o);
	const float u0 = x0 - mx, u1 = u0 + ext;
	const float v0 = y0 - my, v1 = v0 + ext;
	float qmin;
	if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) {
		qmin = 0.f;
	} else {
		const float invA = 1.0f / A, invC = 1.0f / C;
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
	return !(qmin > thr + margin);
}
__device__ __forceinline__ bool quad_keep(const float4 q0, const float4 q1, float x0, float y0) { return rect_keep(q0, q1, x0, y0, 7.0f); }

// Gradient hand-off without global atomics.  Every (tile, Gaussian) instance owns one 48-byte slot
//   [0..2] dL_dcolor  [3..4] sum w*dx, sum w*dy  [5..7] sum w*dx*dx, w*dx*dy, w*dy*dy  [8] sum w  [9..11] unused
// with w = dL_dalpha * G per contributing pixel: components 3..7 are stored WITHOUT their per-Gaussian
// constants (-opacity*W/2, -opacity*H/2 and the conic for the mean2D pair, -opacity/2 for the conic terms):
// preprocess_bwd (partials.h) applies them once per Gaussian instead of once per pixel pair
// in `partials`, indexed by emission order: a Gaussian's instances were emitted contiguously
// (row-major over its tile rectangle, binning.hip), so slot = first_slot + (ty-miny)*w + (tx-minx),
// all of which ride in the spare words of the blend record.  The backward blend merges the four
// quads of a tile in LDS and writes each touched slot once with plain stores; preprocess_bwd then
// sums each Gaussian's contiguous run.  (The reference issues 9 atomics per pixel-Gaussian pair;
// 33 M float atomics per view were measured to cost ~1 ms on MI355X at the C3 size.)
constexpr int GRAD_ACC_FLOATS = 12;

}  // namespace gsr
