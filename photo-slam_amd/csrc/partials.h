// partials.h -- the per-Gaussian end of the atomic-free gradient hand-off (blend.h): sum a Gaussian's contiguous run of
// per-instance gradient slots, written and flagged by the backward blend, in a fixed order (bit-reproducible).
// Called by every lane of the wave from wave-uniform control flow; lane = Gaussian (id order, so the few screen-filling
// splats are spread over many waves).  Runs of up to 64 slots are summed by their owner lane -- 16 flag bytes per load,
// squeezed to a bit mask, then one iteration per TOUCHED slot (about 1 in 5: the rest lie behind their tile's last
// contributor or blend into no pixel); longer runs are summed by the whole wave (strided 48-byte slots, then a DPP
// reduction) so that a 3000-tile splat costs 50 iterations, not 3000.
//   a[0..2] dL_dcolor   a[3], a[4] sum w dx, sum w dy   a[5..7] sum w dx dx, w dx dy, w dy dy   a[8] sum w
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

__device__ __forceinline__ void wave_sum_partial_runs(uint32_t cnt, uint32_t first, const float* __restrict__ partials,
                                                      const uint8_t* __restrict__ touched, float (&a)[9])
{
	const int l = lane_id();
	const float4* part4 = reinterpret_cast<const float4*>(partials);
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
}

}  // namespace gsr
