// shrows.h -- coalesced access to per-Gaussian rows of 16-byte vectors (SH coefficients and their
// gradients: 48 floats = 192 bytes per Gaussian at degree 3), and the SH basis shared by the forward and
// backward per-Gaussian kernels.
//
// One thread per Gaussian reading "its" 192-byte row issues 12 loads whose 64 lanes hit 64
// different cache lines each; with ~20 waves per CU the 12 KiB-per-wave footprint thrashes the
// 32 KiB L1 and lines are re-fetched from L2 (measured: 0.9 TB/s effective in preprocess_fwd).
// Here the wave moves whole rows instead: consecutive lanes move consecutive 16-byte pieces of
// the same row (4 rows of 12 vectors per wave instruction), through a per-wave LDS slice whose
// rows are padded to 13 vectors so that both the row-wise fill and the lane-wise drain are
// bank-conflict free.  Only the rows of lanes that want them are touched (culled Gaussians cost
// no SH traffic in the forward pass).
//
// The slice holds STAGE_ROWS = 32 rows (6.5 KiB per wave), not 64: both kernels are latency-bound and their
// occupancy is set by LDS (measured: halving the resident waves doubles their time), so a wave stages its rows
// in passes of 32 and the owner lanes stream them out of LDS one 16-byte vector at a time -- the 48 coefficients
// never sit in registers together.  (The two translation units that include this file are built with
// -fno-slp-vectorize: the SLP vectoriser pairs the per-coefficient products of different coefficients into
// v_pk_mul_f32 and thereby keeps a whole row live -- 138 instead of 80 VGPRs in the SH backward.)
#pragma once
#include "state.h"
#include "wave64.h"
#include "kernels.h"

namespace gsr {

constexpr int ROW_F4 = 12;        // 48 floats
constexpr int ROW_F4_PAD = 13;    // LDS row pitch in float4
#ifndef GSR_STAGE_ROWS
#define GSR_STAGE_ROWS 32
#endif
constexpr int STAGE_ROWS = GSR_STAGE_ROWS;    // rows staged per pass (16 or 32: the movers take four rows per instruction)

// Both movers give 16 lanes to a row (12 of them active at degree 3): four rows per wave instruction, and every index
// is a shift or a compile-time offset -- the 5-rows-per-instruction packing needed divisions by 12 whose results
// (one set per unrolled step) stayed live across the callers' loops and cost ~40 VGPRs.
//
// Fill s_rows[j][0..nf4) for j = 0 .. count-1 with the first nf4 vectors of row (first_row + s_list[list_first + j])
// (BY_SOURCE: the row lands in s_rows[s_list[..]] instead, i.e. at its owner's position).
// gbase points at row 0; row pitch is ROW_F4 vectors.  count <= STAGE_ROWS.  Wave-uniform arguments.
template <bool BY_SOURCE = false>
__device__ __forceinline__ void wave_load_listed_rows(const float4* __restrict__ gbase, size_t first_row, int nf4, int list_first,
                                                      int count, float4 (*s_rows)[ROW_F4_PAD], const uint32_t* s_list)
{
	const int l = lane_id();
	const int slot = l >> 4, col = l & 15;
	for (int j0 = 0; j0 < count; j0 += 4) {
		const int j = j0 + slot;
		if (col < nf4 && j < count) {
			const uint32_t src = s_list[list_first + j];
			// streaming load: a row is read once per pass over the model (the next reader comes after ~3 GB of other traffic)
			s_rows[BY_SOURCE ? (int)src : j][col] = load_stream_f4(gbase + (first_row + src) * ROW_F4 + col);
		}
	}
	wave_fence();
}

// Write rows [first_row, first_row + nrows) (nrows <= STAGE_ROWS) from s_rows to global: each instruction writes four
// whole rows = 768 contiguous bytes.
__device__ __forceinline__ void wave_store_rows(float4* __restrict__ gbase, size_t first_row, int nrows,
                                                float4 (*s_rows)[ROW_F4_PAD])
{
	const int l = lane_id();
	const int slot = l >> 4, col = l & 15;
	wave_fence();
	float4* dst = gbase + (first_row + slot) * ROW_F4 + col;
	const float4* src = &s_rows[slot][col];
#pragma unroll
	for (int k = 0; k < STAGE_ROWS / 4; k++) {
		if (col < ROW_F4 && 4 * k + slot < nrows) dst[4 * k * ROW_F4] = src[4 * k * ROW_F4_PAD];
	}
	wave_fence();
}

// The update term of one element, m / (sqrt(v) c2 + eps), for the SH rows (eager, culled and lazy paths alike -- they must
// agree bit for bit): v_sqrt_f32 and v_rcp_f32 (1 ulp each) instead of the correctly rounded expansions the compiler emits
// for sqrtf and '/'.  The term is multiplied by a step size of ~1e-3 before it meets a parameter of ~1e-1..1: its relative
// error of 2e-7 is 1e-9 of the parameter, far below the parameter's own rounding (6e-8), and against torch::optim::Adam the
// result stays within one ulp of the parameter (tests/test_train_step.py, tests/test_gpu_parity.py).  It matters because the
// zero-gradient steps of the culled rows run next to other kernels on a second stream: the expansions cost ~90 issue cycles per
// element and step (51 M element-steps per step at C3), this form 31.
__device__ __forceinline__ float adam_ratio(float m, float v, float c2, float eps)
{
#ifdef GSR_EMU
	return m * (1.0f / (sqrtf(v) * c2 + eps));
#else
	return m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * c2 + eps);
#endif
}

// One Adam step (train_ops.hip: adam_kernel, same formula) on rows [first_row, first_row + nrows) whose GRADIENT sits in
// s_rows: the movers read parameter and moments from global (four whole rows = 768 contiguous bytes per instruction and
// array), update them and write them back; the gradient never reaches HBM.  The first 3 floats of a row (features_dc) use
// step_size, the other 45 (features_rest) step_size_tail.
struct RowAdam {
	float* param;
	float* exp_avg;
	float* exp_avg_sq;
	AdamScalars s;
};
// row_mask: bit r clear = row r of the stage is left alone (it took its step elsewhere: gsr_backward, side stream; or it takes
// it later: lazy mode).
// CATCH_UP (lazy mode of gsr_sh_adam_from_views): stage row r lags s_lag[r] steps behind (step - 1); the mover takes those
// zero-gradient steps first -- parameter and moments are in its registers anyway -- and then this step with the gradient.
__device__ __forceinline__ void lazy_zero_grad_steps(const LazyAdamTable& t, int k_hi, int k_lo, int col, float4& pv, float4& mv,
                                                     float4& vv);
template <bool CATCH_UP = false>
__device__ __forceinline__ void wave_adam_rows(const RowAdam& a, size_t first_row, int nrows, float4 (*s_rows)[ROW_F4_PAD],
                                               uint32_t row_mask = 0xFFFFFFFFu, const LazyAdamTable* t = nullptr,
                                               const uint32_t* s_lag = nullptr)
{
	const int l = lane_id();
	const int slot = l >> 4, col = l & 15;
	wave_fence();
	const size_t base = (first_row + slot) * ROW_F4 + col;
	const float ss_first = col == 0 ? a.s.step_size : a.s.step_size_tail;   // .x .y .z of vector 0 are features_dc
#ifndef GSR_ADAM_UNROLL
#define GSR_ADAM_UNROLL 2
#endif
#pragma unroll GSR_ADAM_UNROLL
	for (int k = 0; k < STAGE_ROWS / 4; k++) {
		if (col < ROW_F4 && 4 * k + slot < nrows && ((row_mask >> (4 * k + slot)) & 1u)) {
			const size_t i = base + (size_t)(4 * k * ROW_F4);
			const float4 gv = s_rows[4 * k + slot][col];
			float4 pv = load_stream_f4(reinterpret_cast<const float4*>(a.param) + i);
			float4 mv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg) + i);
			float4 vv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg_sq) + i);
			if (CATCH_UP) {
				const int lag = (int)s_lag[4 * k + slot];
				if (lag > 0) lazy_zero_grad_steps(*t, lag, 1, col, pv, mv, vv);
			}
			float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const float ss = e < 3 ? ss_first : a.s.step_size_tail;
				mp[e] = a.s.b1 * mp[e] + a.s.omb1 * gp[e];
				vp[e] = a.s.b2 * vp[e] + a.s.omb2 * gp[e] * gp[e];
				pp[e] -= ss * adam_ratio(mp[e], vp[e], a.s.inv_sqrt_bc2, a.s.eps);
			}
			store_stream_f4(reinterpret_cast<float4*>(a.param) + i, pv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg) + i, mv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg_sq) + i, vv);
		}
	}
	wave_fence();
}

// The fused step of sh_bwd_rows_kernel: the gradient row of a visible Gaussian is rank one, dL_dsh[k][ch] = basis_k(dir) *
// dRGB[ch] (backward.cu:41-127), and the row's PARAMETER is in the stage already (the SH backward needs the coefficients for the
// view-direction term).  So the stage keeps the parameter, the owner leaves the 16 basis values and the 3 colour gradients in
// s_aux (19 floats instead of a 48-float gradient row written over the parameter), and the movers form their four gradient
// elements as the same products -- the parameter is not read from HBM a second time (192 B per visible Gaussian).
// vis_mask: rows whose parameter and s_aux entry are staged; the other rows of row_mask step with a zero gradient and read
// their parameter from HBM as wave_adam_rows does.
constexpr int AUX_PITCH = 21;   // 16 basis values + 3 colour gradients, odd pitch: conflict-free owner writes
__device__ __forceinline__ void wave_adam_rows_rank1(const RowAdam& a, size_t first_row, int nrows, float4 (*s_rows)[ROW_F4_PAD],
                                                     const float (*s_aux)[AUX_PITCH], uint32_t row_mask, uint32_t vis_mask)
{
	const int l = lane_id();
	const int slot = l >> 4, col = l & 15;
	wave_fence();
	const size_t base = (first_row + slot) * ROW_F4 + col;
	const float ss_first = col == 0 ? a.s.step_size : a.s.step_size_tail;
	int kc[4], cc[4];   // element e = 4 col + c of the row is coefficient e / 3, channel e % 3
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const int e = (4 * col + c) % (4 * ROW_F4);   // (lanes with col >= ROW_F4 do nothing below)
		kc[c] = e / 3;
		cc[c] = 16 + e - 3 * kc[c];
	}
#pragma unroll GSR_ADAM_UNROLL
	for (int k = 0; k < STAGE_ROWS / 4; k++) {
		const int r = 4 * k + slot;
		if (col < ROW_F4 && r < nrows && ((row_mask >> r) & 1u)) {
			const size_t i = base + (size_t)(4 * k * ROW_F4);
			float4 mv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg) + i);
			float4 vv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg_sq) + i);
			float4 pv, gv;
			if ((vis_mask >> r) & 1u) {
				pv = s_rows[r][col];
				const float* ax = s_aux[r];
				gv = make_float4(ax[kc[0]] * ax[cc[0]], ax[kc[1]] * ax[cc[1]], ax[kc[2]] * ax[cc[2]], ax[kc[3]] * ax[cc[3]]);
			} else {
				pv = load_stream_f4(reinterpret_cast<const float4*>(a.param) + i);
				gv = make_float4(0.f, 0.f, 0.f, 0.f);
			}
			float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const float ss = e < 3 ? ss_first : a.s.step_size_tail;
				mp[e] = a.s.b1 * mp[e] + a.s.omb1 * gp[e];
				vp[e] = a.s.b2 * vp[e] + a.s.omb2 * gp[e] * gp[e];
				pp[e] -= ss * adam_ratio(mp[e], vp[e], a.s.inv_sqrt_bc2, a.s.eps);
			}
			store_stream_f4(reinterpret_cast<float4*>(a.param) + i, pv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg) + i, mv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg_sq) + i, vv);
		}
	}
	wave_fence();
}

// ---------------------------------------------------------------------------------------------------------------
// Lazy Adam for rows whose gradient is zero (gsr_sh_adam_lazy, include/gsr.h).  A Gaussian the view culls gets a zero SH
// gradient, and a zero-gradient Adam step of a row depends on nothing but the row itself and the step's scalars:
//   m <- b1 m;  v <- b2 v;  p <- p - step_size m / (sqrt(v) c + eps)
// so it can be taken LATER, several at a time with the row in registers -- the same arithmetic in the same order, one HBM
// round trip instead of one per step.  row_step[i] counts the Adam steps row i has taken; a row is brought up to date
//   * by the forward pass when it becomes visible (before its coefficients are evaluated),
//   * by the backward pass of every `window`-th step (a rotating slice of the row blocks), so that no row lags by more than
//     `window` steps and the table of past scalars stays short,
//   * by gsr_sh_adam_flush (all rows) before anybody else reads the tensor or its moments.
// (LazyAdamTable / LazyAdam: kernels.h)

// The zero-gradient Adam steps (step - k_hi) .. (step - k_lo), oldest first, on one 16-byte vector of a row (col = its index
// in the row: .x .y .z of vector 0 are features_dc).  The expressions are those of wave_adam_rows with g = 0.
__device__ __forceinline__ void lazy_zero_grad_steps(const LazyAdamTable& t, int k_hi, int k_lo, int col, float4& pv, float4& mv,
                                                     float4& vv)
{
	float* pp = &pv.x; float* mp = &mv.x; float* vp = &vv.x;
	for (int k = k_hi; k >= k_lo; k--) {
		const float tail = t.step_size_tail[k];
		const float ss_first = col == 0 ? t.step_size[k] : tail;
		const float c2 = t.inv_sqrt_bc2[k];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const float ss = e < 3 ? ss_first : tail;
			const float g = 0.f;
			mp[e] = t.b1 * mp[e] + t.omb1 * g;
			vp[e] = t.b2 * vp[e] + t.omb2 * g * g;
			pp[e] -= ss * adam_ratio(mp[e], vp[e], c2, t.eps);
		}
	}
}

// Forward pass: rows j = 0 .. count-1 of the stage belong to the lanes s_list[list_first + j] of the wave and lag
// s_lag[list_first + j] steps behind (step - 1); the whole rows (ROW_F4 vectors) sit in s_rows.  The movers bring the lagging
// ones up to date in HBM (parameter and moments) and in the stage.
__device__ __forceinline__ void wave_lazy_catch_up_listed(const LazyAdam& a, size_t first_row, int list_first, int count,
                                                          float4 (*s_rows)[ROW_F4_PAD], const uint32_t* s_list, const uint32_t* s_lag)
{
	const int l = lane_id();
	const int slot = l >> 4, col = l & 15;
	wave_fence();
	for (int j0 = 0; j0 < count; j0 += 4) {
		const int j = j0 + slot;
		if (col < ROW_F4 && j < count) {
			const int lag = (int)s_lag[list_first + j];
			if (lag > 0) {
				const size_t i = (first_row + s_list[list_first + j]) * ROW_F4 + col;
				float4 pv = s_rows[j][col];
				float4 mv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg) + i);
				float4 vv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg_sq) + i);
				lazy_zero_grad_steps(a.t, lag, 1, col, pv, mv, vv);
				store_stream_f4(reinterpret_cast<float4*>(a.param) + i, pv);
				store_stream_f4(reinterpret_cast<float4*>(a.exp_avg) + i, mv);
				store_stream_f4(reinterpret_cast<float4*>(a.exp_avg_sq) + i, vv);
				s_rows[j][col] = pv;
			}
		}
	}
	wave_fence();
}

// ---------------------------------------------------------------------------------------------------------------
// SH basis, cuda_rasterizer/auxiliary.h:22-39 + forward.cu:20-71 / backward.cu:20-139.
__device__ static const float SHB_C0 = 0.28209479177387814f;
__device__ static const float SHB_C1 = 0.4886025119029199f;
__device__ static const float SHB_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SHB_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

struct ShDir {
	float x, y, z, xx, yy, zz, xy, yz, xz;
};
__device__ __forceinline__ ShDir sh_dir(float x, float y, float z)
{
	ShDir d;
	d.x = x; d.y = y; d.z = z;
	d.xx = x * x; d.yy = y * y; d.zz = z * z;
	d.xy = x * y; d.yz = y * z; d.xz = x * z;
	return d;
}

// Basis value of coefficient k, written with the operand order of the reference so that (basis * sh) summed in
// coefficient order reproduces computeColorFromSH bit for bit under -ffp-contract=off (the signs of degree 1 are
// folded into the basis: (-a)*b == -(a*b) and r + (-t) == r - t exactly).  k is a compile-time constant after
// unrolling.
__device__ __forceinline__ float sh_basis(int k, const ShDir& d)
{
	switch (k) {
	case 0: return SHB_C0;
	case 1: return -SHB_C1 * d.y;
	case 2: return SHB_C1 * d.z;
	case 3: return -SHB_C1 * d.x;
	case 4: return SHB_C2[0] * d.xy;
	case 5: return SHB_C2[1] * d.yz;
	case 6: return SHB_C2[2] * (2.0f * d.zz - d.xx - d.yy);
	case 7: return SHB_C2[3] * d.xz;
	case 8: return SHB_C2[4] * (d.xx - d.yy);
	case 9: return SHB_C3[0] * d.y * (3.0f * d.xx - d.yy);
	case 10: return SHB_C3[1] * d.xy * d.z;
	case 11: return SHB_C3[2] * d.y * (4.0f * d.zz - d.xx - d.yy);
	case 12: return SHB_C3[3] * d.z * (2.0f * d.zz - 3.0f * d.xx - 3.0f * d.yy);
	case 13: return SHB_C3[4] * d.x * (4.0f * d.zz - d.xx - d.yy);
	case 14: return SHB_C3[5] * d.z * (d.xx - d.yy);
	default: return SHB_C3[6] * d.x * (d.xx - 3.0f * d.yy);
	}
}

// d(basis k)/d(x, y, z) of the unit direction, backward.cu:63-127 (the factors that multiply sh[k] in dRGBdx/dy/dz).
// Returns which of the three are non-zero (bit 0: x, 1: y, 2: z) so that the caller adds nothing for the others.
__device__ __forceinline__ int sh_basis_grad(int k, const ShDir& d, float& gx, float& gy, float& gz)
{
	gx = gy = gz = 0.f;
	switch (k) {
	case 0: return 0;
	case 1: gy = -SHB_C1; return 2;
	case 2: gz = SHB_C1; return 4;
	case 3: gx = -SHB_C1; return 1;
	case 4: gx = SHB_C2[0] * d.y; gy = SHB_C2[0] * d.x; return 3;
	case 5: gy = SHB_C2[1] * d.z; gz = SHB_C2[1] * d.y; return 6;
	case 6: gx = SHB_C2[2] * 2.f * -d.x; gy = SHB_C2[2] * 2.f * -d.y; gz = SHB_C2[2] * 2.f * 2.f * d.z; return 7;
	case 7: gx = SHB_C2[3] * d.z; gz = SHB_C2[3] * d.x; return 5;
	case 8: gx = SHB_C2[4] * 2.f * d.x; gy = SHB_C2[4] * 2.f * -d.y; return 3;
	case 9: gx = SHB_C3[0] * 3.f * 2.f * d.xy; gy = SHB_C3[0] * 3.f * (d.xx - d.yy); return 3;
	case 10: gx = SHB_C3[1] * d.yz; gy = SHB_C3[1] * d.xz; gz = SHB_C3[1] * d.xy; return 7;
	case 11: gx = SHB_C3[2] * -2.f * d.xy; gy = SHB_C3[2] * (-3.f * d.yy + 4.f * d.zz - d.xx); gz = SHB_C3[2] * 4.f * 2.f * d.yz; return 7;
	case 12: gx = SHB_C3[3] * -3.f * 2.f * d.xz; gy = SHB_C3[3] * -3.f * 2.f * d.yz; gz = SHB_C3[3] * 3.f * (2.f * d.zz - d.xx - d.yy); return 7;
	case 13: gx = SHB_C3[4] * (-3.f * d.xx + 4.f * d.zz - d.yy); gy = SHB_C3[4] * -2.f * d.xy; gz = SHB_C3[4] * 4.f * 2.f * d.xz; return 7;
	case 14: gx = SHB_C3[5] * 2.f * d.xz; gy = SHB_C3[5] * -2.f * d.yz; gz = SHB_C3[5] * (d.xx - d.yy); return 7;
	default: gx = SHB_C3[6] * 3.f * (d.xx - d.yy); gy = SHB_C3[6] * -3.f * 2.f * d.xy; return 3;
	}
}

// computeColorFromSH before the +0.5 / clamp: streams the row one vector at a time (ncoef = (deg+1)^2, wave-uniform).
__device__ __forceinline__ void sh_row_to_rgb(const float4* row, int ncoef, const ShDir& d, float (&rgb)[3])
{
	rgb[0] = rgb[1] = rgb[2] = 0.f;
#pragma unroll
	for (int i = 0; i < ROW_F4; i++) {
		if ((4 * i) / 3 < ncoef) {
			const float4 v = row[i];
			const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const int j = 4 * i + c, k = j / 3, ch = j - 3 * k;
				if (k < ncoef) {
					const float t = sh_basis(k, d) * in[c];
					rgb[ch] = (k == 0) ? t : rgb[ch] + t;
				}
			}
		}
	}
}

// SH backward in place: the row holds sh on entry and (WRITE) dL_dsh = basis * dRGB on exit (zeros beyond ncoef);
// dd{x,y,z}[ch] accumulate dRGB/d(direction).
template <bool WRITE = true>
__device__ __forceinline__ void sh_row_backward(float4* row, int ncoef, const ShDir& d, const float (&dRGB)[3], float (&ddx)[3],
                                                float (&ddy)[3], float (&ddz)[3])
{
#pragma unroll
	for (int i = 0; i < ROW_F4; i++) {
		float out[4] = {0.f, 0.f, 0.f, 0.f};
		if ((4 * i) / 3 < ncoef) {
			const float4 v = row[i];
			const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const int j = 4 * i + c, k = j / 3, ch = j - 3 * k;
				if (k < ncoef) {
					out[c] = sh_basis(k, d) * dRGB[ch];
					float gx, gy, gz;
					const int nz = sh_basis_grad(k, d, gx, gy, gz);
					if (nz & 1) ddx[ch] += gx * in[c];
					if (nz & 2) ddy[ch] += gy * in[c];
					if (nz & 4) ddz[ch] += gz * in[c];
				}
			}
		}
		if (WRITE) row[i] = make_float4(out[0], out[1], out[2], out[3]);
	}
}

}  // namespace gsr
