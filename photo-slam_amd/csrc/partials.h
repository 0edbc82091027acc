// partials.h -- the per-Gaussian end of the atomic-free gradient hand-off (blend.h): sum a Gaussian's contiguous run of
// per-instance gradient slots, written and flagged by the backward blend, in a fixed order (bit-reproducible).
// Lane = Gaussian.  Runs of up to LONG_RUN (64) slots are summed by their owner lane -- 16 flag bytes per load, squeezed to a
// bit mask, then one iteration per TOUCHED slot (about 1 in 5: the rest lie behind their tile's last contributor or blend
// into no pixel).  Longer runs (screen-filling splats; listed by the forward preprocess) are summed beforehand by
// long_run_sums_kernel, one WAVE per run (strided 48-byte slots, then a DPP reduction: a 3000-tile splat costs 50 iterations),
// which leaves the total in the run's first slot: wherever such Gaussians sit in the arrays -- densification appends the
// children of split Gaussians consecutively -- no wave of the backward preprocess inherits their work.
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
	if (cnt != 0u && cnt <= LONG_RUN) {
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
	// a longer run was summed by long_run_sums_kernel (one wave per run), which left the total in the run's FIRST slot
	if (cnt > LONG_RUN && touched[first]) {
		const float4 x = part4[3 * (size_t)first], y = part4[3 * (size_t)first + 1];
		const float z = part4[3 * (size_t)first + 2].x;
		a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w;
		a[4] = y.x; a[5] = y.y; a[6] = y.z; a[7] = y.w;
		a[8] = z;
	}
}

// One wave per listed run: sum its touched slots in a fixed order, store the total in the run's first slot and flag it.
__device__ __forceinline__ void wave_sum_long_run(uint32_t first, uint32_t cnt, float* __restrict__ partials, uint8_t* __restrict__ touched)
{
	const int l = lane_id();
	float4* part4 = reinterpret_cast<float4*>(partials);
	float v[9];
#pragma unroll
	for (int c = 0; c < 9; c++) v[c] = 0.f;
	const float4* src = part4 + 3 * (size_t)first;
	bool any = false;
	// four slots per lane and trip: the flags first, then the touched slots' loads together, then the sums in slot order -- the
	// kernel's time is the longest run's chain of dependent (flag, slot) round trips
	constexpr int U = 4;
	for (uint32_t i0 = (uint32_t)l; i0 < cnt; i0 += 64u * U) {
		bool t[U];
		float4 x[U], y[U];
		float z[U];
#pragma unroll
		for (int j = 0; j < U; j++) {
			const uint32_t i = i0 + 64u * (uint32_t)j;
			t[j] = i < cnt && touched[first + i] != 0;
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			const size_t i = (size_t)i0 + 64u * (size_t)j;
			if (t[j]) {
				x[j] = src[3 * i];
				y[j] = src[3 * i + 1];
				z[j] = src[3 * i + 2].x;
			}
		}
#pragma unroll
		for (int j = 0; j < U; j++) {
			if (t[j]) {
				any = true;
				v[0] += x[j].x; v[1] += x[j].y; v[2] += x[j].z; v[3] += x[j].w;
				v[4] += y[j].x; v[5] += y[j].y; v[6] += y[j].z; v[7] += y[j].w;
				v[8] += z[j];
			}
		}
	}
	const bool some = wave_ballot(any) != 0ull;
	wave_reduce9_f32(v);  // totals in lane 63; every lane has read its slots by now (the reduction is a rendezvous)
	if (l == 63) {
		part4[3 * (size_t)first] = make_float4(v[0], v[1], v[2], v[3]);
		part4[3 * (size_t)first + 1] = make_float4(v[4], v[5], v[6], v[7]);
		reinterpret_cast<float*>(part4 + 3 * (size_t)first + 2)[0] = v[8];
		touched[first] = some ? 1 : 0;
	}
}

}  // namespace gsr
