// partials.h -- the per-Gaussian end of the atomic-free gradient hand-off (blend.h): sum a Gaussian's contiguous run of
// per-instance gradient slots, written and flagged by the backward blend, in a fixed order (bit-reproducible).
// Lane = Gaussian.  Runs of up to LONG_RUN (64) slots are summed by their owner lane -- 16 flag bytes per load, squeezed to a
// bit mask, then the TOUCHED slots only (about 1 in 5: the rest lie behind their tile's last contributor or blend into no
// pixel), two per trip.  Longer runs (screen-filling splats; listed by the forward pass) are summed beforehand by
// long_run_sums_kernel, one WAVE per run, which leaves the total in the run's first slot: wherever such Gaussians sit in the
// arrays -- densification appends the children of split Gaussians consecutively -- no wave of the backward preprocess inherits
// their work.
//   a[0..2] dL_dcolor   a[3], a[4] sum w dx, sum w dy   a[5..7] sum w dx dx, w dx dy, w dy dy   a[8] sum w
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

// U touched slots of a run in one trip: the loads first, the sums in slot order
template <int U>
__device__ __forceinline__ void sum_slots_trip(const float4* src, unsigned long long& live, float (&a)[9])
{
	int idx[U];
	float4 x[U], y[U];
	float z[U];
#pragma unroll
	for (int j = 0; j < U; j++) {
		idx[j] = __ffsll((long long)live) - 1;
		live &= live - 1ull;
	}
#pragma unroll
	for (int j = 0; j < U; j++) {
		x[j] = src[SLOT_F4 * (size_t)idx[j]];
		y[j] = src[SLOT_F4 * (size_t)idx[j] + 1];
		z[j] = src[SLOT_F4 * (size_t)idx[j] + 2].x;
	}
#pragma unroll
	for (int j = 0; j < U; j++) {
		a[0] += x[j].x; a[1] += x[j].y; a[2] += x[j].z; a[3] += x[j].w;
		a[4] += y[j].x; a[5] += y[j].y; a[6] += y[j].z; a[7] += y[j].w;
		a[8] += z[j];
	}
}

// 16 flag bytes (0 / 1 each) at p, squeezed to 16 bits (bit i = byte i)
__device__ __forceinline__ uint32_t squeeze_flags16(const uint8_t* p)
{
	uint4 f;
	__builtin_memcpy(&f, p, 16);   // unaligned 16-byte load
	return (((f.x * 0x01020408u) >> 24) & 0xFu) | ((((f.y * 0x01020408u) >> 24) & 0xFu) << 4) | ((((f.z * 0x01020408u) >> 24) & 0xFu) << 8) |
	       ((((f.w * 0x01020408u) >> 24) & 0xFu) << 12);
}

__device__ __forceinline__ void wave_sum_partial_runs(uint32_t cnt, uint32_t first, const float* __restrict__ partials,
                                                      const uint8_t* __restrict__ touched, float (&a)[9])
{
	const float4* part4 = reinterpret_cast<const float4*>(partials);
#pragma unroll
	for (int c = 0; c < 9; c++) a[c] = 0.f;
	if (cnt != 0u && cnt <= LONG_RUN) {
		// the run's flags, 16 bytes per load, squeezed to one bit per slot: the loops below then run over the TOUCHED slots
		// only (~1 in 5) and their loads do not wait for one another (a byte-flag test per slot serialises on memory latency)
		unsigned long long live = 0ull;
#pragma unroll
		for (int c = 0; c < 4; c++)
			if (16u * c < cnt) live |= (unsigned long long)squeeze_flags16(touched + first + 16 * c) << (16 * c);
		if (cnt < 64u) live &= (1ull << cnt) - 1ull;
		const float4* src = part4 + SLOT_F4 * (size_t)first;
		// two touched slots per trip: their loads leave together, the sums follow in slot order (a lane's chain of dependent round
		// trips is what this HBM-latency-bound phase waits for: one slot per trip -> two: the stage 0.463 -> 0.452 ms at C3; four: equal)
		while (live & (live - 1ull)) sum_slots_trip<2>(src, live, a);
		while (live) sum_slots_trip<1>(src, live, a);
	}
	// a longer run was summed by long_run_sums_kernel (one wave per run), which left the total in the run's FIRST slot
	if (cnt > LONG_RUN && touched[first]) {
		const float4 x = part4[SLOT_F4 * (size_t)first], y = part4[SLOT_F4 * (size_t)first + 1];
		const float z = part4[SLOT_F4 * (size_t)first + 2].x;
		a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w;
		a[4] = y.x; a[5] = y.y; a[6] = y.z; a[7] = y.w;
		a[8] = z;
	}
}

// One wave per listed run: sum its touched slots in a fixed order, store the total in the run's first slot and flag it.  Lane l owns
// the k = ceil(cnt / 64) CONSECUTIVE slots [l k, (l + 1) k): 16-byte flag loads, then only the touched slots, four per trip.
// (Rounds 2-5 also carried two lane-strided forms behind GSR_LRS_MODE; this one measured fastest -- 20 -> 14 us at C3 -- and stayed.)
__device__ __forceinline__ void wave_sum_long_run(uint32_t first, uint32_t cnt, float* __restrict__ partials, uint8_t* __restrict__ touched)
{
	const int l = lane_id();
	float4* part4 = reinterpret_cast<float4*>(partials);
	float v[9];
#pragma unroll
	for (int c = 0; c < 9; c++) v[c] = 0.f;
	bool any = false;
	const uint32_t k = (cnt + 63u) >> 6;
	const uint32_t lo = (uint32_t)l * k, hi = min(cnt, lo + k);
	for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {   // (one pass for runs of up to 4 096 slots)
		const uint32_t n = min(64u, hi - g0);
		unsigned long long live = 0ull;
#pragma unroll
		for (int c = 0; c < 4; c++)
			if (16u * c < n) live |= (unsigned long long)squeeze_flags16(touched + first + g0 + 16 * c) << (16 * c);   // (the array is padded by 64 bytes)
		if (n < 64u) live &= (1ull << n) - 1ull;
		any = any || live != 0ull;
		const float4* src = part4 + SLOT_F4 * (size_t)(first + g0);
		while (__popcll(live) >= 4) sum_slots_trip<4>(src, live, v);
		while (live & (live - 1ull)) sum_slots_trip<2>(src, live, v);
		while (live) sum_slots_trip<1>(src, live, v);
	}
	const bool some = wave_ballot(any) != 0ull;
	wave_reduce9_f32(v);  // totals in lane 63; every lane has read its slots by now (the reduction is a rendezvous)
	if (l == 63) {
		part4[SLOT_F4 * (size_t)first] = make_float4(v[0], v[1], v[2], v[3]);
		part4[SLOT_F4 * (size_t)first + 1] = make_float4(v[4], v[5], v[6], v[7]);
		reinterpret_cast<float*>(part4 + SLOT_F4 * (size_t)first + 2)[0] = v[8];
		touched[first] = some ? 1 : 0;
	}
}

}  // namespace gsr
