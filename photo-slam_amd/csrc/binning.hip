// binning.hip -- instance generation and per-tile ranges.
//
// The reference builds 64-bit (tile | depth) keys for every (tile, Gaussian) instance and
// radix-sorts all R of them on 32+msb(T) bits (rasterizer_impl.cu:70-111, 303-308).  Here the
// LSD sort is split at the 32-bit boundary: the P Gaussians are sorted once by depth bits
// (stable, so ties keep ascending id), instances are then emitted in that order, and only
// the tile-id digits are sorted over the R instances.  An LSD radix sort is a sequence of
// stable passes from the least significant digit up, and emitting a Gaussian's instances
// contiguously commutes with the low-digit passes (all its instances share the depth bits),
// so the final order is identical to the reference's: (tile, depth bits, Gaussian id,
// row-major tile order) -- but 4 of the 6 passes run over P elements instead of R.
#include "state.h"
#include "wave64.h"
#include "kernels.h"

namespace gsr {

// One wave emits the instances of 64 depth-consecutive Gaussians, lane-consecutively, so the
// key/value stores are fully coalesced and a screen-filling splat is spread over all lanes
// (duplicateWithKeys gives each Gaussian's whole run to one thread).
__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                      const uint32_t* __restrict__ tiles_touched, const uint16_t* __restrict__ rect, int grid_x,
                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
	__shared__ uint32_t s_off[4][64];
	__shared__ uint32_t s_g[4][64];
	__shared__ uint2 s_rect[4][64];
	const int w = wave_id(), l = lane_id();
	const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const bool valid = j < P;
	const uint32_t g = valid ? order[j] : 0u;
	const uint32_t cnt = valid ? tiles_touched[g] : 0u;
	const uint32_t off = valid ? offsets[j] : 0u;
	const uint32_t total = wave_sum_u32(cnt);
	if (total == 0) return;  // wave-uniform
	// wave base = offset of the first valid lane = min over lanes holding instances; offsets are
	// monotone in j, invalid lanes sit at the end, so lane 0 always holds the base.
	const uint32_t base = wave_shfl_u32(off, 0);
	s_off[w][l] = valid ? off - base : 0xFFFFFFFFu;
	s_g[w][l] = g;
	s_rect[w][l] = valid ? reinterpret_cast<const uint2*>(rect)[g] : make_uint2(0u, 0u);
	wave_fence();
	for (uint32_t i = (uint32_t)l; i < total; i += 64u) {
		// last lane whose (relative, exclusive) offset is <= i; zero-count lanes share their
		// successor's offset, so "last" skips them.
		int lo = 0;
#pragma unroll
		for (int step = 32; step >= 1; step >>= 1)
			if (s_off[w][lo + step] <= i) lo += step;
		const uint32_t k = i - s_off[w][lo];
		const uint2 r = s_rect[w][lo];
		const uint32_t minx = r.x & 0xFFFFu, miny = r.x >> 16, maxx = r.y & 0xFFFFu;
		const uint32_t wdt = maxx - minx;
		const uint32_t yy = k / wdt;
		const uint32_t xx = k - yy * wdt;
		keys[base + i] = (miny + yy) * (uint32_t)grid_x + (minx + xx);
		vals[base + i] = s_g[w][lo];
	}
}

// identifyTileRanges, rasterizer_impl.cu:116-138, on 32-bit tile keys.
__global__ void __launch_bounds__(256)
tile_ranges_kernel(int R, const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= R) return;
	const uint32_t cur = tile_keys[i];
	if (i == 0)
		ranges[cur].x = 0;
	else {
		const uint32_t prev = tile_keys[i - 1];
		if (cur != prev) {
			ranges[prev].y = (uint32_t)i;
			ranges[cur].x = (uint32_t)i;
		}
	}
	if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

int launch_emit_instances(int P, const GeometryState& g, int grid_x, uint32_t* keys, uint32_t* vals, hipStream_t stream)
{
	GSR_LAUNCH(emit_instances_kernel, div_up(P, 256), 256, stream, P, (const uint32_t*)g.order, (const uint32_t*)g.offsets,
	           (const uint32_t*)g.tiles_touched, (const uint16_t*)g.rect, grid_x, keys, vals);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_tile_ranges(int R, const uint32_t* tile_keys, uint2* ranges, hipStream_t stream)
{
	if (R <= 0) return GSR_OK;
	GSR_LAUNCH(tile_ranges_kernel, div_up(R, 256), 256, stream, R, tile_keys, ranges);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
