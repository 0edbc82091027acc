// shrows.h -- coalesced access to per-Gaussian rows of 16-byte vectors (SH coefficients and their
// gradients: 48 floats = 192 bytes per Gaussian at degree 3).
//
// One thread per Gaussian reading "its" 192-byte row issues 12 loads whose 64 lanes hit 64
// different cache lines each; with ~20 waves per CU the 12 KiB-per-wave footprint thrashes the
// 32 KiB L1 and lines are re-fetched from L2 (measured: 0.9 TB/s effective in preprocess_fwd).
// Here the wave moves whole rows instead: consecutive lanes move consecutive 16-byte pieces of
// the same row (5 rows of 12 vectors per wave instruction), through a per-wave LDS slice whose
// rows are padded to 13 vectors so that both the row-wise fill and the lane-wise drain are
// bank-conflict free.  Only the rows of lanes that want them are touched (culled Gaussians cost
// no SH traffic in the forward pass).
#pragma once
#include "state.h"
#include "wave64.h"

namespace gsr {

constexpr int ROW_F4 = 12;        // 48 floats
constexpr int ROW_F4_PAD = 13;    // LDS row pitch in float4

// Fill s_rows[lane][0..nf4) with the first nf4 vectors of row (first_row + lane) for every lane with
// `want`.  gbase points at row 0; row pitch is ROW_F4 vectors.  s_list is a 64-entry scratch.
__device__ __forceinline__ void wave_load_rows(const float4* __restrict__ gbase, size_t first_row, int nf4, bool want,
                                               float4 (*s_rows)[ROW_F4_PAD], uint32_t* s_list)
{
	const int l = lane_id();
	const unsigned long long mask = wave_ballot(want);
	const int nvis = __popcll(mask);
	if (nvis == 0) return;  // wave-uniform
	if (want) s_list[__popcll(mask & lanemask_lt())] = (uint32_t)l;
	wave_fence();
	const int per = 64 / nf4;          // rows per wave instruction
	const int slot = l / nf4, col = l - slot * nf4;
	for (int it = 0; it * per < nvis; it++) {
		const int r = it * per + slot;
		if (slot < per && r < nvis) {
			const uint32_t src = s_list[r];
			s_rows[src][col] = gbase[(first_row + src) * ROW_F4 + col];
		}
	}
	wave_fence();
}

// Write rows [first_row, first_row + nrows) (nrows <= 64) from s_rows to global, fully coalesced.
__device__ __forceinline__ void wave_store_rows(float4* __restrict__ gbase, size_t first_row, int nrows,
                                                float4 (*s_rows)[ROW_F4_PAD])
{
	const int l = lane_id();
	wave_fence();
#pragma unroll
	for (int k = 0; k < ROW_F4; k++) {
		const int i = l + 64 * k;
		const int row = i / ROW_F4, col = i - row * ROW_F4;
		if (row < nrows) gbase[first_row * ROW_F4 + i] = s_rows[row][col];
	}
	wave_fence();
}

}  // namespace gsr
