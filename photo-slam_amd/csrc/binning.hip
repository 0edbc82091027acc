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

// Instance emission, load-balanced over SLOTS (duplicateWithKeys, rasterizer_impl.cu:70-111, gives each Gaussian's
// whole run to one thread).  Every wave owns EMIT_SLOTS consecutive instance slots, whatever Gaussians they
// belong to: a 64-ary search over the depth-ordered exclusive offsets (four dependent, fully parallel probes
// for 2 M Gaussians) finds the Gaussian holding the wave's first slot, the next EMIT_SLOTS + 1 offsets go to LDS
// (every visible Gaussian owns >= 1 slot, so the window always suffices; culled ones sort to the end with
// offset == R and are never selected), and each lane locates its slot's Gaussian with an 8-step LDS binary
// search.  Screen-filling splats near the camera are consecutive in depth order; a per-Gaussian decomposition
// left a single wave with >100 k instances of them (the kernel's tail was 80 % of its time).
constexpr int EMIT_SLOTS = 256;

__global__ void __launch_bounds__(256)
emit_instances_kernel(int P, uint32_t R, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                      const uint16_t* __restrict__ rect, int grid_x, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                      float4* __restrict__ rec)
{
	__shared__ uint32_t s_off[4][EMIT_SLOTS + 4];
	const int w = wave_id(), l = lane_id();
	const uint32_t s0 = ((uint32_t)blockIdx.x * 4u + (uint32_t)w) * (uint32_t)EMIT_SLOTS;
	if (s0 >= R) return;   // wave-uniform
	const uint32_t n = (R - s0) < (uint32_t)EMIT_SLOTS ? (R - s0) : (uint32_t)EMIT_SLOTS;
	// r0 = last depth rank whose offset is <= s0 (offsets[0] == 0 keeps the invariant offsets[lo] <= s0)
	uint32_t lo = 0, hi = (uint32_t)P;
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
		const uint2 r = reinterpret_cast<const uint2*>(rect)[g];
		const uint32_t minx = r.x & 0xFFFFu, miny = r.x >> 16, maxx = r.y & 0xFFFFu;
		const uint32_t wdt = maxx - minx;
		const uint32_t yy = k / wdt;
		const uint32_t xx = k - yy * wdt;
		keys[slot] = (miny + yy) * (uint32_t)grid_x + (minx + xx);
		vals[slot] = g;
		// slot of the Gaussian's first instance = its emission offset (the backward blend writes its per-tile gradient
		// partials there, reduce_partials sums the contiguous run)
		if (k == 0u) reinterpret_cast<uint32_t*>(rec + 3 * (size_t)g + 2)[3] = slot;
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

// Sum every Gaussian's contiguous run of per-instance gradient slots (written by the backward
// blend) into one 48-byte record per Gaussian, in a fixed order (no atomics: bit-reproducible).
// Thread = Gaussian (id order, so the few screen-filling splats are spread over many waves).
// Runs of up to 64 slots are summed by their owner lane; longer runs are summed by the whole wave
// (strided 48-byte slots, then a DPP reduction) so that a 3000-tile splat costs 50 iterations,
// not 3000.
__global__ void __launch_bounds__(256)
reduce_partials_kernel(int P, const float4* __restrict__ rec, const uint32_t* __restrict__ tiles_touched,
                       const float* __restrict__ partials, const uint8_t* __restrict__ touched, float* __restrict__ grad_acc,
                       float half_w, float half_h)
{
	const int l = lane_id();
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const bool valid = idx < P;
	const uint32_t cnt = valid ? tiles_touched[idx] : 0u;
	if (wave_ballot(cnt != 0u) == 0ull) return;  // wave-uniform
	const uint32_t first = cnt ? __float_as_uint(rec[3 * (size_t)idx + 2].w) : 0u;
	const float4* part4 = reinterpret_cast<const float4*>(partials);
	float a[9];
#pragma unroll
	for (int c = 0; c < 9; c++) a[c] = 0.f;
	if (cnt != 0u && cnt <= 64u) {
		// the run's flags, 16 bytes per load, squeezed to one bit per slot: the loop below then runs once per TOUCHED
		// slot (~1 in 5) and its loads do not wait for one another (a byte-flag test per slot serialises on memory latency)
		unsigned long long live = 0ull;
#pragma unroll
		for (int c = 0; c < 4; c++) {
			if (16u * c < cnt) {
				uint4 f;
				__builtin_memcpy(&f, touched + first + 16 * c, 16);   // unaligned 16-byte load
				const uint32_t bits = (((f.x * 0x01020408u) >> 24) & 0xFu) | ((((f.y * 0x01020408u) >> 24) & 0xFu) << 4) |
				                      ((((f.z * 0x01020408u) >> 24) & 0xFu) << 8) | ((((f.w * 0x01020408u) >> 24) & 0xFu) << 12);
				live |= (unsigned long long)bits << (16 * c);
			}
		}
		if (cnt < 64u) live &= (1ull << cnt) - 1ull;
		const float4* src = part4 + 3 * (size_t)first;
		while (live) {
			const int i = __ffsll((long long)live) - 1;
			live &= live - 1ull;
			const float4 x = src[3 * (size_t)i], y = src[3 * (size_t)i + 1];
			const float z = src[3 * (size_t)i + 2].x;
			a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
			a[4] += y.x; a[5] += y.y; a[6] += y.z; a[7] += y.w;
			a[8] += z;
		}
	}
	unsigned long long big = wave_ballot(cnt > 64u);
	while (big) {
		const int b = __ffsll((long long)big) - 1;
		big &= big - 1ull;
		const uint32_t bfirst = wave_readlane_u32(first, b), bcnt = wave_readlane_u32(cnt, b);
		float v[9];
#pragma unroll
		for (int c = 0; c < 9; c++) v[c] = 0.f;
		const float4* src = part4 + 3 * (size_t)bfirst;
		for (uint32_t i = (uint32_t)l; i < bcnt; i += 64u) {
			if (!touched[bfirst + i]) continue;
			const float4 x = src[3 * (size_t)i], y = src[3 * (size_t)i + 1];
			const float z = src[3 * (size_t)i + 2].x;
			v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
			v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
			v[8] += z;
		}
		wave_reduce9_f32(v);  // totals in lane 63
#pragma unroll
		for (int c = 0; c < 9; c++) a[c] = wave_writelane_f32(a[c], wave_readlane_f32(v[c], 63), b);
	}
	if (cnt) {
		// constant factors the blend left out: dL_dG = opacity * dL_dalpha; d(mean2D) carries -W/2, -H/2
		// (backward.cu:460-461,539-546), the conic terms -1/2 (:549-551)
		// a[3], a[4] = sum dL_dG*G*dx, sum dL_dG*G*dy: dG_ddelx = -G (dx A + dy B), dG_ddely = -G (dy C + dx B)
		const float4 q0 = rec[3 * (size_t)idx];
		const float4 q1 = rec[3 * (size_t)idx + 1];
		const float A = q0.z, B = q0.w, C = q1.x, o = q1.y;
		float4* dst = reinterpret_cast<float4*>(grad_acc) + 3 * (size_t)idx;
		dst[0] = make_float4(a[0], a[1], a[2], -o * half_w * (a[3] * A + a[4] * B));
		dst[1] = make_float4(-o * half_h * (a[4] * C + a[3] * B), -0.5f * o * a[5], -0.5f * o * a[6], -0.5f * o * a[7]);
		dst[2] = make_float4(a[8], 0.f, 0.f, 0.f);
	}
}

int launch_reduce_partials(int P, const GeometryState& g, const float* partials, const uint8_t* touched, float* grad_acc, int W, int H,
                           hipStream_t stream)
{
	GSR_LAUNCH(reduce_partials_kernel, div_up(P, 256), 256, stream, P, (const float4*)g.rec, (const uint32_t*)g.tiles_touched,
	           partials, touched, grad_acc, 0.5f * (float)W, 0.5f * (float)H);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_emit_instances(int P, int R, const GeometryState& g, int grid_x, uint32_t* keys, uint32_t* vals, hipStream_t stream)
{
	if (R <= 0) return GSR_OK;
	GSR_LAUNCH(emit_instances_kernel, div_up(R, 4 * EMIT_SLOTS), 256, stream, P, (uint32_t)R, (const uint32_t*)g.order,
	           (const uint32_t*)g.offsets, (const uint16_t*)g.rect, grid_x, keys, vals, g.rec);
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
