// partials.h -- the per-Gaussian end of the atomic-free gradient hand-off (blend.h): sum a Gaussian's contiguous run of
// per-instance gradient slots, written and flagged by the backward blend, in a fixed order.
// Lane = Gaussian.  Runs of up to LONG_RUN (64) slots are summed by their owner lane -- 16 flag bytes per load, squeezed to a
// bit mask, then one iteration per TOUCHED slot (about 1 in 5: the rest lie behind their tile's last contributor or blend
// into no pixel).  A longer run (a screen-filling splat) arrives FOLDED into its first LONG_FOLD slots (state.h: the backward
// blend added instance k's sums to slot k % LONG_FOLD with float atomics): the owner lane sums those and leaves them zeroed for
// the next backward pass.  Wherever such Gaussians sit in the arrays -- densification appends the children of split Gaussians
// consecutively -- no lane sums more than LONG_RUN slots.
//   a[0..2] dL_dcolor   a[3], a[4] sum w dx, sum w dy   a[5..7] sum w dx dx, w dx dy, w dy dy   a[8] sum w
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

// U touched slots of a run in one trip: the loads first, the sums in slot order; a folded run's accumulators are zeroed again
template <int U>
__device__ __forceinline__ void sum_slots_trip(float4* src, unsigned long long& live, bool folded, float (&a)[9])
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
	if (folded) {
		const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int j = 0; j < U; j++) {
			src[SLOT_F4 * (size_t)idx[j]] = zero;
			src[SLOT_F4 * (size_t)idx[j] + 1] = zero;
			src[SLOT_F4 * (size_t)idx[j] + 2] = zero;
		}
	}
}

__device__ __forceinline__ void wave_sum_partial_runs(uint32_t cnt, uint32_t first, float* __restrict__ partials,
                                                      const uint8_t* __restrict__ touched, uint32_t fold, float (&a)[9], int trip = 2)
{
	const int l = lane_id();
	float4* part4 = reinterpret_cast<float4*>(partials);
#pragma unroll
	for (int c = 0; c < 9; c++) a[c] = 0.f;
	const bool folded = cnt > LONG_RUN;
	const uint32_t n = folded ? fold : cnt;   // slots that hold this Gaussian's sums
	if (n != 0u) {
		// the run's flags, 16 bytes per load, squeezed to one bit per slot: the loop below then runs once per TOUCHED
		// slot (~1 in 5) and its loads do not wait for one another (a byte-flag test per slot serialises on memory latency)
		unsigned long long live = 0ull;
#pragma unroll
		for (int c = 0; c < 4; c++) {
			if (16u * c < n) {
				uint4 f;
				__builtin_memcpy(&f, touched + first + 16 * c, 16);   // unaligned 16-byte load
				const uint32_t bits = (((f.x * 0x01020408u) >> 24) & 0xFu) | ((((f.y * 0x01020408u) >> 24) & 0xFu) << 4) |
				                      ((((f.z * 0x01020408u) >> 24) & 0xFu) << 8) | ((((f.w * 0x01020408u) >> 24) & 0xFu) << 12);
				live |= (unsigned long long)bits << (16 * c);
			}
		}
		if (n < 64u) live &= (1ull << n) - 1ull;
		float4* src = part4 + SLOT_F4 * (size_t)first;
		// several touched slots per trip: their loads leave together, the sums follow in slot order (a lane's chain of dependent
		// round trips is what this HBM-latency-bound phase waits for: one slot per trip -> two: the stage 0.463 -> 0.452 ms at C3)
		if (trip >= 4)
			while (__popcll(live) >= 4) sum_slots_trip<4>(src, live, folded, a);
		if (trip >= 2)
			while (live & (live - 1ull)) sum_slots_trip<2>(src, live, folded, a);
		while (live) {
			const int i = __ffsll((long long)live) - 1;
			live &= live - 1ull;
			const float4 x = src[SLOT_F4 * (size_t)i], y = src[SLOT_F4 * (size_t)i + 1];
			const float z = src[SLOT_F4 * (size_t)i + 2].x;
			a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
			a[4] += y.x; a[5] += y.y; a[6] += y.z; a[7] += y.w;
			a[8] += z;
			if (folded) {   // an accumulator of the atomics of blend_bwd: zero again for the next backward pass
				const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
				src[SLOT_F4 * (size_t)i] = zero;
				src[SLOT_F4 * (size_t)i + 1] = zero;
				src[SLOT_F4 * (size_t)i + 2] = zero;
			}
		}
	}
}

}  // namespace gsr
