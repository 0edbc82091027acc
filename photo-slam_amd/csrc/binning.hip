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
#include "blend.h"

namespace gsr {

// Instance emission, load-balanced over SLOTS (duplicateWithKeys, rasterizer_impl.cu:70-111, gives each Gaussian's
// whole run to one thread).  Every wave owns EMIT_SLOTS consecutive instance slots, whatever Gaussians they
// belong to: a 64-ary search over the depth-ordered exclusive offsets (four dependent, fully parallel probes
// for 2 M Gaussians) finds the Gaussian holding the wave's first slot, the next EMIT_SLOTS + 1 offsets go to LDS
// (every visible Gaussian owns >= 1 slot, so the window always suffices; culled ones sort to the end with
// offset == R and are never selected), and each lane locates its slot's Gaussian with an 8-step LDS binary
// search.  Screen-filling splats near the camera are consecutive in depth order; a per-Gaussian decomposition
// left a single wave with >100 k instances of them (the kernel's tail was 80 % of its time).
constexpr int EMIT_SLOTS = (int)EMIT_SEED_STRIDE;
constexpr int EMIT_WAVES = SORT_CHUNK / EMIT_SLOTS, EMIT_THREADS = 64 * EMIT_WAVES;   // a workgroup emits one chunk of the tile sort
static_assert(EMIT_WAVES * EMIT_SLOTS == SORT_CHUNK && EMIT_THREADS <= 1024, "one emission workgroup per sort chunk");

__global__ void __launch_bounds__(EMIT_THREADS)
emit_instances_kernel(int P, uint32_t R, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                      const uint2* __restrict__ rect_sorted, int grid_x, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                      float4* __restrict__ rec, uint8_t* __restrict__ touched, uint32_t touched_bytes, int cull,
                      const uint32_t* __restrict__ seeds, uint32_t seed_capacity, uint32_t* __restrict__ hist, int hist_bits, int hist_blocks)
{
	__shared__ uint32_t s_off[EMIT_WAVES][EMIT_SLOTS + 4];
	// The tile sort's first histogram (sort.hip: radix_hist_kernel) is counted HERE, where the keys are made: a workgroup emits
	// exactly one sort chunk (EMIT_WAVES x EMIT_SLOTS == SORT_CHUNK slots), counts its keys' low digits in LDS and stores its
	// column of the [digit][block] table -- one launch and one pass over the 25 MB of keys less per forward pass.
	__shared__ uint32_t s_hist[1 << RADIX_BITS_ONE_PASS];   // (up to 11 bits: the one-pass tile sort of a small view, state.h)
	if (hist) {
		for (int i = (int)threadIdx.x; i < (1 << hist_bits); i += EMIT_THREADS) s_hist[i] = 0u;
		__syncthreads();
	}
	const int w = wave_id(), l = lane_id();
	const uint32_t s0 = ((uint32_t)blockIdx.x * (uint32_t)EMIT_WAVES + (uint32_t)w) * (uint32_t)EMIT_SLOTS;
	const bool idle = s0 >= R;   // wave-uniform (the waves behind the last slot: they still meet the others at the barrier below)
	if (!idle) {
	const uint32_t n = (R - s0) < (uint32_t)EMIT_SLOTS ? (R - s0) : (uint32_t)EMIT_SLOTS;
	// The slot flags of the backward blend (state.h: touched) start out cleared: this wave clears those of its slots (the wave
	// that holds the last slot: up to the end of the padded array) -- the backward pass then needs no memset of its own in
	// front of the blend (two 5 us fill kernels and the bubble behind them); it leaves the flags cleared again when it is done
	// (sh_bwd_rows_kernel), so any number of backward passes may follow one forward pass.
	{
		const uint32_t c_end = (s0 + (uint32_t)EMIT_SLOTS >= R) ? touched_bytes : s0 + (uint32_t)EMIT_SLOTS;
		for (uint32_t o = s0 + 4u * (uint32_t)l; o < c_end; o += 256u) *reinterpret_cast<uint32_t*>(touched + o) = 0u;
	}
	// r0 = last depth rank whose offset is <= s0 (offsets[0] == 0 keeps the invariant offsets[lo] <= s0): left by the offset
	// scan for this window (one load instead of four dependent probes of the offsets per wave), searched for only beyond the
	// seed table's capacity
	const uint32_t window = s0 / (uint32_t)EMIT_SLOTS;
	const bool seeded = seeds != nullptr && window < seed_capacity;   // wave-uniform
	uint32_t lo = seeded ? wave_uniform_u32(seeds[window]) : 0u, hi = seeded ? lo + 1u : (uint32_t)P;
	while (hi - lo > 1u) {
		const uint32_t step = (hi - lo + 63u) >> 6;
		const uint32_t probe = lo + (uint32_t)l * step;
		const bool le = probe < hi && offsets[probe] <= s0;
		const uint32_t c = (uint32_t)__popcll(wave_ballot(le));   // monotone: lanes 0 .. c-1
		lo = wave_uniform_u32(lo + (c - 1u) * step);
		hi = wave_uniform_u32(min(hi, lo + step));
	}
	const uint32_t r0 = lo;
	for (uint32_t i = (uint32_t)l; i <= (uint32_t)EMIT_SLOTS; i += 64u)
		s_off[w][i] = (i <= n && r0 + i < (uint32_t)P) ? offsets[r0 + i] : 0xFFFFFFFFu;
	wave_fence();
	for (uint32_t i = (uint32_t)l; i < n; i += 64u) {
		const uint32_t slot = s0 + i;
		uint32_t j = 0;   // last window entry whose offset is <= slot
#pragma unroll
		for (uint32_t step = EMIT_SLOTS / 2; step >= 1u; step >>= 1)
			if (s_off[w][j + step] <= slot) j += step;
		const uint32_t k = slot - s_off[w][j];
		const uint32_t g = order[r0 + j];
		const uint2 r = rect_sorted[r0 + j];   // (in depth order: left there by the offset scan, sort.hip)
		const uint32_t minx = r.x & 0xFFFFu, miny = r.x >> 16, maxx = r.y & 0xFFFFu;
		const uint32_t wdt = maxx - minx;
		const uint32_t yy = k / wdt;
		const uint32_t xx = k - yy * wdt;
		uint32_t key = (miny + yy) * (uint32_t)grid_x + (minx + xx);
		// GSR_CULL_EMPTY_TILES: the instance of a tile in which no pixel can blend this Gaussian (the blend kernels' own
		// conservative test, on the tile's 16 x 16 rectangle) gets the key that the tile sort's first pass drops
		if (cull && !rect_keep(rec[3 * (size_t)g], rec[3 * (size_t)g + 1], (float)((minx + xx) * TILE), (float)((miny + yy) * TILE), (float)(TILE - 1)))
			key = RADIX_INVALID_KEY;
		keys[slot] = key;
		vals[slot] = g;
		// slot of the Gaussian's first instance = its emission offset (the backward blend writes its per-tile gradient
		// partials there, preprocess_bwd sums the contiguous run)
		if (k == 0u) reinterpret_cast<uint32_t*>(rec + 3 * (size_t)g + 2)[3] = slot;
		// (one LDS atomic per key, as in radix_hist_kernel; a key the tile sort's first pass drops is not counted)
		if (hist && key != RADIX_INVALID_KEY) atomicAdd(&s_hist[key & ((1u << hist_bits) - 1u)], 1u);
	}
	}   // !idle
	if (hist) {
		__syncthreads();
		for (int d = (int)threadIdx.x; d < (1 << hist_bits); d += EMIT_THREADS) hist[(size_t)d * hist_blocks + blockIdx.x] = s_hist[d];
	}
}

// identifyTileRanges, rasterizer_impl.cu:116-138, on 32-bit tile keys.  Four consecutive keys per thread (one 16-byte load + the
// key in front of them): a quarter of the threads and loads of the one-key-per-thread form for the same 25 MB.
__global__ void __launch_bounds__(256)
tile_ranges_kernel(int R, const uint32_t* __restrict__ tile_keys, uint2* __restrict__ ranges, const uint32_t* __restrict__ n_dev)
{
	if (n_dev) R = min(R, (int)*n_dev);   // (the list was compacted by the tile sort: GSR_CULL_EMPTY_TILES)
	const int i0 = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i0 >= R) return;
	uint32_t k[4];
	if (i0 + 3 < R) {
		const uint4 v = *reinterpret_cast<const uint4*>(tile_keys + i0);   // (the key arrays are 16-byte aligned: state.h)
		k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
	} else {
#pragma unroll
		for (int j = 0; j < 4; j++) k[j] = i0 + j < R ? tile_keys[i0 + j] : 0u;
	}
	uint32_t prev = i0 > 0 ? tile_keys[i0 - 1] : 0u;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int i = i0 + j;
		if (i >= R) break;
		const uint32_t cur = k[j];
		if (i == 0)
			ranges[cur].x = 0;
		else if (cur != prev) {
			ranges[prev].y = (uint32_t)i;
			ranges[cur].x = (uint32_t)i;
		}
		if (i == R - 1) ranges[cur].y = (uint32_t)R;
		prev = cur;
	}
}

int launch_emit_instances(int P, int R, const GeometryState& g, int grid_x, uint32_t* keys, uint32_t* vals, uint8_t* touched,
                          hipStream_t stream, int cull, bool seeded, uint32_t* hist, int hist_bits)
{
	if (R <= 0) return GSR_OK;
	// hist (nullable): the [digit][sort_blocks(R)] table of the tile sort's first pass over `hist_bits` low key bits, counted on the way
	GSR_LAUNCH(emit_instances_kernel, div_up(R, EMIT_WAVES * EMIT_SLOTS), EMIT_THREADS, stream, P, (uint32_t)R, (const uint32_t*)g.order,
	           (const uint32_t*)g.offsets, (const uint2*)g.rect_sorted, grid_x, keys, vals, g.rec, touched,
	           (uint32_t)touched_clear_bytes((size_t)R), cull, seeded ? (const uint32_t*)g.sort_keys_b : (const uint32_t*)nullptr, (uint32_t)P,
	           hist_bits > 0 ? hist : (uint32_t*)nullptr, hist_bits, sort_blocks(R));
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_tile_ranges(int R, const uint32_t* tile_keys, uint2* ranges, hipStream_t stream, const uint32_t* n_dev)
{
	if (R <= 0) return GSR_OK;
	GSR_LAUNCH(tile_ranges_kernel, div_up(R, 4 * 256), 256, stream, R, tile_keys, ranges, n_dev);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
