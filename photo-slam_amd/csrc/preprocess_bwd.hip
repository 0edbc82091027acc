// preprocess_bwd.hip -- per-Gaussian backward stage.
//
// Two kernels for the reference model's layout (48-float, 16-byte aligned SH rows):
//   * preprocess_bwd_kernel: sums the Gaussian's per-instance gradient slots (the per-Gaussian end of the atomic-free
//     hand-off from the backward blend, partials.h) in registers, then computeCov2DCUDA (backward.cu:144-274) fused with
//     the projection and cov3D backward of preprocessCUDA (backward.cu:346-396, :278-341): neither the blend-stage
//     gradients nor dL_dcov3D nor the partial dL_dmean3D round-trip through HBM between launches.  It writes every output
//     element but dL_dsh (zeros for culled Gaussians), so the caller does not need the reference's torch::zeros pass
//     (src/rasterize_points.cu:149-157, 300 B/Gaussian);
//   * sh_bwd_rows_kernel: the SH backward (cuda_rasterizer/backward.cu:20-139).  Rows move through LDS (shrows.h); it
//     writes dL_dsh for EVERY Gaussian and adds the view-direction term to dL_dmean3D last (the reference's order).
// They were one kernel until its 140+ VGPRs capped it at three waves per SIMD; both halves are
// latency-bound (measured: time ~ 1/occupancy), and apart they run at 6-8 waves.  Other SH layouts take the
// fused instantiation with per-lane row access.
//
// HBM per Gaussian: culled: 4 read (radius) + (55+3M)*4 written zeros; visible: reads mean 12 (x2), cov3D 24, record 32,
// 36 per touched instance slot + 1 flag byte per slot, SH 12K, scale 12, rot 16, clamp 1; writes mean3D 12 (+12 re-read
// and rewritten by the SH kernel), mean2D 12, colour 12 (re-read once), opacity 4, cov3D 24, SH 12M, scale 12, rot 16.
#include "state.h"
#include "wave64.h"
#include <stdlib.h>
#include "kernels.h"
#include "shrows.h"
#include "partials.h"

namespace gsr {


constexpr int PRB_THREADS = 128;
constexpr int SHB_THREADS = 64;    // one wave x 6.5 KiB of row staging per workgroup

// One Adam step of row idx of a [P,N] geometry tensor with the gradient in registers (gsr_geom_adam): the thread that holds
// the gradient applies it -- the gradient is not written and not read back, the four separate passes disappear.
template <int N>
__device__ __forceinline__ void geom_adam_row(const GeomAdamTensor& t, size_t idx, const float (&g)[N])
{
#pragma unroll
	for (int k = 0; k < N; k++) {
		const size_t i = (size_t)N * idx + k;
		const float m = t.s.b1 * t.exp_avg[i] + t.s.omb1 * g[k];
		const float v = t.s.b2 * t.exp_avg_sq[i] + t.s.omb2 * g[k] * g[k];
		t.exp_avg[i] = m;
		t.exp_avg_sq[i] = v;
		t.param[i] -= t.s.step_size * adam_ratio(m, v, t.s.inv_sqrt_bc2, t.s.eps);
	}
}
__device__ __forceinline__ void geom_adam_row4(const GeomAdamTensor& t, size_t idx, const float4& g4)
{
	float4 pv = reinterpret_cast<const float4*>(t.param)[idx];
	float4 mv = reinterpret_cast<const float4*>(t.exp_avg)[idx];
	float4 vv = reinterpret_cast<const float4*>(t.exp_avg_sq)[idx];
	float* pp = &pv.x; float* mp = &mv.x; float* vp = &vv.x;
	const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
	for (int k = 0; k < 4; k++) {
		mp[k] = t.s.b1 * mp[k] + t.s.omb1 * g[k];
		vp[k] = t.s.b2 * vp[k] + t.s.omb2 * g[k] * g[k];
		pp[k] -= t.s.step_size * adam_ratio(mp[k], vp[k], t.s.inv_sqrt_bc2, t.s.eps);
	}
	reinterpret_cast<float4*>(t.param)[idx] = pv;
	reinterpret_cast<float4*>(t.exp_avg)[idx] = mv;
	reinterpret_cast<float4*>(t.exp_avg_sq)[idx] = vv;
}

// SH backward for aligned 48-float rows; DEG = active SH degree.  FACTORED (gsr_backward_args.dL_dcolor_view): the
// gradient rows are not produced -- the clamp-masked colour gradient leaves instead (12 B instead of 192 B per Gaussian)
// and gsr_sh_grad_from_views rebuilds dL_dsh for the whole keyframe batch after the exchange.
template <int DEG, int MODE>   // MODE: 0 = gradient rows out, 1 = factored (colour gradient out), 2 = fused Adam step
#ifndef GSR_SHB_WAVES_LO
#define GSR_SHB_WAVES_LO 6
#endif
__global__ void __launch_bounds__(SHB_THREADS) GSR_WAVES_PER_EU(GSR_SHB_WAVES_LO, 8)   // 80 VGPRs (3 dwords of scratch at degree 3)
sh_bwd_rows_kernel(const PreprocessBwdParams p)
{
	__shared__ float4 s_rows[SHB_THREADS / 64][STAGE_ROWS][ROW_F4_PAD];
	__shared__ uint32_t s_list[SHB_THREADS / 64][STAGE_ROWS];
	// fused step (MODE 2): the stage keeps the PARAMETER rows; basis values + colour gradient of each row (shrows.h: rank one)
	__shared__ float s_aux[SHB_THREADS / 64][MODE == 2 ? STAGE_ROWS : 1][AUX_PITCH];
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int w = wave_id(), l = lane_id();
	const size_t wave_first = (size_t)(blockIdx.x * blockDim.x) + (size_t)w * 64;
	// the slot flags have been consumed by the kernels in front of this one: cleared here for the next backward pass (the
	// forward pass hands them over cleared: emit_instances_kernel), 16 bytes per thread
	if (p.touched_clear)
		for (uint32_t o = 16u * (uint32_t)idx; o < p.touched_clear_bytes; o += 16u * (uint32_t)(gridDim.x * blockDim.x))
			*reinterpret_cast<uint4*>(p.touched_clear + o) = make_uint4(0u, 0u, 0u, 0u);
	const bool in_range = idx < p.P;
	const bool vis = in_range && (p.radii[idx] > 0);
	constexpr int ncoef = (DEG + 1) * (DEG + 1);
	constexpr bool FACTORED = MODE == 1;
	float dRGB[3] = {0.f, 0.f, 0.f};
	float ddx[3] = {0.f, 0.f, 0.f}, ddy[3] = {0.f, 0.f, 0.f}, ddz[3] = {0.f, 0.f, 0.f};  // dRGBdx/dy/dz
	float ox = 0.f, oy = 0.f, oz = 1.f;
	if (vis) {
		ox = p.means3D[3 * (size_t)idx] - p.campos[0];
		oy = p.means3D[3 * (size_t)idx + 1] - p.campos[1];
		oz = p.means3D[3 * (size_t)idx + 2] - p.campos[2];
		if (FACTORED) {
			// the clamp-masked colour gradient was written by preprocess_bwd_kernel, which runs first -- it is the quantity the
			// exchange gathers, complete one kernel earlier than this one (a visible Gaussian with an all-zero gradient carries
			// -0.0f in channel 0 there: numerically nothing)
			dRGB[0] = p.dL_dcolor_view[3 * (size_t)idx + 0];
			dRGB[1] = p.dL_dcolor_view[3 * (size_t)idx + 1];
			dRGB[2] = p.dL_dcolor_view[3 * (size_t)idx + 2];
		} else {
			const uint8_t cl = p.clamped[idx];   // dL_dcolor: written by preprocess_bwd_kernel, which runs first
			dRGB[0] = p.dL_dcolor[3 * (size_t)idx + 0] * ((cl & 1) ? 0.f : 1.f);
			dRGB[1] = p.dL_dcolor[3 * (size_t)idx + 1] * ((cl & 2) ? 0.f : 1.f);
			dRGB[2] = p.dL_dcolor[3 * (size_t)idx + 2] * ((cl & 4) ? 0.f : 1.f);
		}
	}
	// The wave handles its 64 rows in two halves of STAGE_ROWS: the SH rows of the half's visible lanes are fetched
	// by the whole wave into LDS, each owner turns its row into the gradient row IN PLACE (zeros for culled
	// Gaussians) while accumulating dRGB/d(direction), and the half leaves as one contiguous 6 KiB burst.
	{
		{
			const int nf4 = MODE == 2 ? ROW_F4 : (3 * ncoef + 3) >> 2;   // (fused step: the movers need the whole parameter row)
#pragma unroll 1
			for (int h = 0; h < 64 / STAGE_ROWS; h++) {
				const size_t half_first = wave_first + (size_t)(h * STAGE_ROWS);
				const long long left = (long long)p.P - (long long)half_first;
				if (left <= 0) break;   // wave-uniform
				const bool mine = (l / STAGE_ROWS) == h;
				const unsigned long long m = wave_ballot(vis && mine);
				if (vis && mine) s_list[w][__popcll(m & lanemask_lt())] = (uint32_t)(l % STAGE_ROWS);
				wave_fence();
				wave_load_listed_rows<true>(reinterpret_cast<const float4*>(p.shs), half_first, nf4, 0, __popcll(m), s_rows[w], s_list[w]);
				if (mine && in_range) {
					float4* row = s_rows[w][l % STAGE_ROWS];
					if (vis) {
						// the direction is made opaque per pass: otherwise the ~50 basis / derivative factors are hoisted out
						// of the loop as invariants and stay live across it
						float ux = ox, uy = oy, uz = oz;
						GSR_OPAQUE_F32(ux);
						GSR_OPAQUE_F32(uy);
						GSR_OPAQUE_F32(uz);
						const float len = sqrtf(ux * ux + uy * uy + uz * uz);
						const ShDir d = sh_dir(ux / len, uy / len, uz / len);
						sh_row_backward<MODE == 0>(row, ncoef, d, dRGB, ddx, ddy, ddz);
						if (MODE == 2) {
							float* ax = s_aux[w][l % STAGE_ROWS];
#pragma unroll
							for (int k = 0; k < 16; k++) ax[k] = k < ncoef ? sh_basis(k, d) : 0.f;
							ax[16] = dRGB[0];
							ax[17] = dRGB[1];
							ax[18] = dRGB[2];
						}
					} else if (MODE == 0) {
#pragma unroll
						for (int i = 0; i < ROW_F4; i++) row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
					}
				}
				if (MODE == 0) {
					wave_store_rows(reinterpret_cast<float4*>(p.dL_dsh), half_first, (int)(left > STAGE_ROWS ? STAGE_ROWS : left), s_rows[w]);
				} else if (MODE == 2) {
					const RowAdam ra = {p.adam_param, p.adam_exp_avg, p.adam_exp_avg_sq, p.adam};
					// the rows of this half that step here: all of them, or only the visible ones when the culled Gaussians
					// took the step in gsr_forward (their rows then are neither read nor written)
					const uint32_t lit = (uint32_t)(m >> (h * STAGE_ROWS));
					const uint32_t rows = p.adam_skip_culled ? lit : 0xFFFFFFFFu;
					wave_adam_rows_rank1(ra, half_first, (int)(left > STAGE_ROWS ? STAGE_ROWS : left), s_rows[w], s_aux[w], rows, lit);
				} else {
					wave_fence();   // the next pass refills the slice
				}
			}
		}
	}
	if (MODE == 2 && vis && p.lazy_row_step) p.lazy_row_step[idx] = p.lazy_step;   // lazy mode: this row has taken the step
	float gm[3] = {0.f, 0.f, 0.f};
	if (vis) {
		const float dLx = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
		const float dLy = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
		const float dLz = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
		// dnormvdv, auxiliary.h:107-117, added to the covariance + projection terms preprocess_bwd_kernel left in
		// dL_dmean3D (the reference's order of the three contributions, backward.cu:271-273,384,137)
		const float sum2 = ox * ox + oy * oy + oz * oz;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		gm[0] = p.dL_dmean3D[3 * (size_t)idx + 0] + ((+sum2 - ox * ox) * dLx - oy * ox * dLy - oz * ox * dLz) * invsum32;
		gm[1] = p.dL_dmean3D[3 * (size_t)idx + 1] + (-ox * oy * dLx + (sum2 - oy * oy) * dLy - oz * oy * dLz) * invsum32;
		gm[2] = p.dL_dmean3D[3 * (size_t)idx + 2] + (-ox * oz * dLx - oy * oz * dLy + (sum2 - oz * oz) * dLz) * invsum32;
		if (!p.geom.on) {
			p.dL_dmean3D[3 * (size_t)idx + 0] = gm[0];
			p.dL_dmean3D[3 * (size_t)idx + 1] = gm[1];
			p.dL_dmean3D[3 * (size_t)idx + 2] = gm[2];
		}
	}
	// the position's gradient is complete here: its Adam step (zero gradient for culled Gaussians, as a dense optimizer)
	if (p.geom.on && in_range) geom_adam_row<3>(p.geom.xyz, (size_t)idx, gm);
}

// The row of Gaussian idx of the view's packed message (include/gsr.h: gsr_backward_args.packed_view).  seen_mask = the ballot of
// "visible" over the wave = the 64-row group idx >> 6; the rows in front of the group come from the prefix section that
// gsr_pack_view_plan computed from the forward pass's radii.  The last group's first lane writes the header.
__device__ __forceinline__ void pack_view_row(const PreprocessBwdParams& p, int idx, bool vis, unsigned long long seen_mask, float v0, float v1,
                                              float v2)
{
	const size_t groups = ((size_t)p.P + 63) >> 6;
	const size_t prefix_words = (groups + 3) & ~(size_t)3, mask_words = (2 * groups + 3) & ~(size_t)3;
	uint32_t* msg = p.packed_msg;
	const size_t group = (size_t)idx >> 6;
	const uint32_t first = msg[PACK_HEADER + group];
	if (vis) {
		const uint32_t rank = first + (uint32_t)__popcll(seen_mask & lanemask_lt());
		if (rank < (uint32_t)p.packed_capacity) {
			float* rows = reinterpret_cast<float*>(msg + PACK_HEADER + prefix_words + mask_words);
			rows[3 * (size_t)rank] = v0;
			rows[3 * (size_t)rank + 1] = v1;
			rows[3 * (size_t)rank + 2] = v2;
		}
	}
	if (idx == p.P - 1) {   // (the last Gaussian: its group knows the total)
		const uint32_t K = first + (uint32_t)__popcll(seen_mask);
		msg[0] = K;
		msg[1] = (uint32_t)p.P;
		msg[2] = (uint32_t)p.packed_capacity;
		msg[3] = K > (uint32_t)p.packed_capacity ? 1u : 0u;
		msg[4] = __float_as_uint(p.campos[0]);
		msg[5] = __float_as_uint(p.campos[1]);
		msg[6] = __float_as_uint(p.campos[2]);
		msg[7] = 0u;
	}
}

// ROWS_OK: dL_dsh / shs rows are 48 floats and 16-byte aligned (the layout of the reference model): sh_bwd_rows_kernel
// follows and adds the SH term; otherwise the SH backward happens here with per-lane scalar row access.
// PACKED: the view's packed message is written next to dL_dcolor_view (gsr_backward_args.packed_view) -- an instantiation of its
// own: as a run-time branch it cost the kernel of the single-GPU step 8 us (181 -> 189 us at C3, profiles/r04_v) with the option off.
template <bool ROWS_OK, bool PACKED = false>
__global__ void __launch_bounds__(PRB_THREADS)
preprocess_bwd_kernel(const PreprocessBwdParams p)
{

	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const bool in_range = idx < p.P;
	const bool vis = in_range && (p.radii[idx] > 0);
	const int M3 = 3 * p.M;
	constexpr bool rows_ok = ROWS_OK;
	// packed view (kernel-uniform): the lanes of this wave are one 64-row group of the message
	const unsigned long long seen_mask = PACKED ? wave_ballot(vis) : 0ull;

	// ------------------------------------------------------------------ gradients of the blend stage (partials.h)
	// the per-instance slots of this Gaussian's tiles are summed here, in registers: colour 0..2, mean2D moments 3..4,
	// conic moments 5..7, opacity 8 never round-trip through HBM
	float a[9];
	{
		const uint32_t cnt = (vis && p.partials) ? p.tiles_touched[idx] : 0u;
		const uint32_t first = cnt ? __float_as_uint(p.rec[3 * (size_t)idx + 2].w) : 0u;
		wave_sum_partial_runs(cnt, first, p.partials, p.touched, a);   // every lane of the wave takes part
	}
	float* out_sh = (p.dL_dsh && !p.dL_dcolor_view && !p.adam_exp_avg && in_range) ? p.dL_dsh + (size_t)idx * M3 : nullptr;

	// Culled Gaussians take the same store instructions as visible ones, with zeros (the reference leaves the
	// torch::zeros content): every output leaves the wave as full contiguous lines.  Separate zero / value stores,
	// each under its own half-empty lane mask, write every line twice partially (measured: 1.6x write traffic).
	if (!rows_ok && out_sh && !vis)
		for (int i = 0; i < M3; i++) out_sh[i] = 0.f;

	const float* V = p.view;
	const float* Pm = p.proj;
	float mx = 0.f, my = 0.f, mz = 0.f;
	float gmx = 0.f, gmy = 0.f, gmz = 0.f;
	float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	// dL_dcolor (x, y, z) + dL_dmean2D.x (w) ; dL_dmean2D.y (x) + dL_dconic (y, z, w)
	float4 ga0 = make_float4(0.f, 0.f, 0.f, 0.f), ga1 = make_float4(0.f, 0.f, 0.f, 0.f);
	float g_opacity = 0.f;

	if (vis) {
		mx = p.means3D[3 * (size_t)idx];
		my = p.means3D[3 * (size_t)idx + 1];
		mz = p.means3D[3 * (size_t)idx + 2];
		// constant factors the blend left out: dL_dG = opacity * dL_dalpha; d(mean2D) carries -W/2, -H/2 and the conic
		// (dG_ddelx = -G (dx A + dy B), dG_ddely = -G (dy C + dx B), backward.cu:460-461,539-546), the conic terms -1/2 (:549-551)
		const float4 q0 = p.rec[3 * (size_t)idx];
		const float4 q1 = p.rec[3 * (size_t)idx + 1];
		const float A = q0.z, B = q0.w, C = q1.x, o = q1.y;
		ga0 = make_float4(a[0], a[1], a[2], -o * p.half_w * (a[3] * A + a[4] * B));
		ga1 = make_float4(-o * p.half_h * (a[4] * C + a[3] * B), -0.5f * o * a[5], -0.5f * o * a[6], -0.5f * o * a[7]);
		g_opacity = a[8];
		// raw logit: d sigmoid = o (1 - o), o = the activated opacity kept in the blend record
		if (p.raw_params & GSR_RAW_OPACITY) g_opacity = g_opacity * o * (1.0f - o);
	}
	const float gcx = ga1.y, gcy = ga1.z, gcz = ga1.w;
	if (in_range) {
		if (p.dL_dmean2D) {
			p.dL_dmean2D[3 * (size_t)idx + 0] = ga0.w;
			p.dL_dmean2D[3 * (size_t)idx + 1] = ga1.x;
			p.dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
		}
		p.dL_dcolor[3 * (size_t)idx + 0] = ga0.x;
		p.dL_dcolor[3 * (size_t)idx + 1] = ga0.y;
		p.dL_dcolor[3 * (size_t)idx + 2] = ga0.z;
		if (rows_ok && p.dL_dcolor_view) {
			// view-factored mode: the clamp-masked colour gradient (backward.cu:41-48) leaves HERE, one kernel before the SH
			// backward that consumes it too -- the exchange's gather can start while sh_bwd_rows_kernel runs.  A VISIBLE Gaussian
			// whose masked gradient is all zero (drawn, but blended into no pixel that matters) leaves -0.0f in channel 0:
			// numerically nothing, but it tells the lazy rows of gsr_sh_adam_from_views that some view SEES this Gaussian -- its
			// row steps now (with a zero gradient) instead of being caught up by the next forward pass of this view
			float v0 = 0.f, v1 = 0.f, v2 = 0.f;
			if (vis) {
				const uint8_t cl = p.clamped[idx];
				v0 = ga0.x * ((cl & 1) ? 0.f : 1.f);
				v1 = ga0.y * ((cl & 2) ? 0.f : 1.f);
				v2 = ga0.z * ((cl & 4) ? 0.f : 1.f);
				if (v0 == 0.f && v1 == 0.f && v2 == 0.f) v0 = -0.0f;
			}
			p.dL_dcolor_view[3 * (size_t)idx + 0] = v0;
			p.dL_dcolor_view[3 * (size_t)idx + 1] = v1;
			p.dL_dcolor_view[3 * (size_t)idx + 2] = v2;
			if (PACKED) pack_view_row(p, idx, vis, seen_mask, v0, v1, v2);
		}
		if (p.geom.on) {
			const float go[1] = {g_opacity};
			geom_adam_row<1>(p.geom.opacity, (size_t)idx, go);
		} else {
			p.dL_dopacity[idx] = g_opacity;
		}
		if (p.dL_dconic) reinterpret_cast<float4*>(p.dL_dconic)[idx] = make_float4(gcx, gcy, 0.f, gcz);
	}
	if (p.stat_accum && vis) {
		// densify_stats_kernel (train_ops.hip; gaussian_mapper.cpp:714-719, gaussian_model.cpp:817-831) on the values at hand
		p.stat_accum[idx] += sqrtf(ga0.w * ga0.w + ga1.x * ga1.x);
		p.stat_denom[idx] += 1.0f;
		p.stat_max_radii[idx] = fmaxf(p.stat_max_radii[idx], (float)p.radii[idx]);
	}
	float shx = 0.f, shy = 0.f, shz = 0.f;   // d(loss)/d(mean) through the view direction of the SH colour

	// ------------------------------------------------------------------ SH backward, backward.cu:20-139
	if (p.shs) {   // wave-uniform
		if (!rows_ok) {   // (aligned 48-float rows: sh_bwd_rows_kernel runs after this kernel and adds its term)
			const int ncoef = (p.D + 1) * (p.D + 1);
			float dRGB[3] = {0.f, 0.f, 0.f};
			float ddx[3] = {0.f, 0.f, 0.f}, ddy[3] = {0.f, 0.f, 0.f}, ddz[3] = {0.f, 0.f, 0.f};  // dRGBdx/dy/dz
			if (vis) {
				const float ox = mx - p.campos[0], oy = my - p.campos[1], oz = mz - p.campos[2];
				const uint8_t cl = p.clamped[idx];
				dRGB[0] = ga0.x * ((cl & 1) ? 0.f : 1.f);
				dRGB[1] = ga0.y * ((cl & 2) ? 0.f : 1.f);
				dRGB[2] = ga0.z * ((cl & 4) ? 0.f : 1.f);
				const float len = sqrtf(ox * ox + oy * oy + oz * oz);
				const ShDir d = sh_dir(ox / len, oy / len, oz / len);
				const float* shrow = p.shs + (size_t)idx * M3;
#pragma unroll
				for (int k = 0; k < 16; k++) {
					if (k < ncoef && k < p.M) {
						float gx, gy, gz;
						const int nz = sh_basis_grad(k, d, gx, gy, gz);
#pragma unroll
						for (int ch = 0; ch < 3; ch++) {
							const float v = shrow[3 * k + ch];
							if (out_sh) out_sh[3 * k + ch] = sh_basis(k, d) * dRGB[ch];
							if (nz & 1) ddx[ch] += gx * v;
							if (nz & 2) ddy[ch] += gy * v;
							if (nz & 4) ddz[ch] += gz * v;
						}
					}
				}
				if (out_sh)
					for (int i = 3 * (ncoef < p.M ? ncoef : p.M); i < M3; i++) out_sh[i] = 0.f;   // keep the row fully written
				const float dLx = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
				const float dLy = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
				const float dLz = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
				// dnormvdv, auxiliary.h:107-117
				const float sum2 = ox * ox + oy * oy + oz * oz;
				const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
				shx = ((+sum2 - ox * ox) * dLx - oy * ox * dLy - oz * ox * dLz) * invsum32;
				shy = (-ox * oy * dLx + (sum2 - oy * oy) * dLy - oz * oy * dLz) * invsum32;
				shz = (-ox * oz * dLx - oy * oz * dLy + (sum2 - oz * oz) * dLz) * invsum32;
			}
			if (p.dL_dcolor_view && in_range) {
				if (vis && dRGB[0] == 0.f && dRGB[1] == 0.f && dRGB[2] == 0.f) dRGB[0] = -0.0f;   // (visibility marker: sh_bwd_rows_kernel)
				p.dL_dcolor_view[3 * (size_t)idx + 0] = dRGB[0];
				p.dL_dcolor_view[3 * (size_t)idx + 1] = dRGB[1];
				p.dL_dcolor_view[3 * (size_t)idx + 2] = dRGB[2];
				if (PACKED) pack_view_row(p, idx, vis, seen_mask, dRGB[0], dRGB[1], dRGB[2]);
			}
		}
	}

	if (vis) {
		// ------------------------------------------------------------------ computeCov2DCUDA, backward.cu:144-274
		float c3[6];
		if (p.cov3D) {   // a covariance the caller precomputed (cov3D_precomp)
#pragma unroll
			for (int i = 0; i < 6; i++) c3[i] = p.cov3D[6 * (size_t)idx + i];
		} else {         // the forward preprocess's own: recomputed, the same bits (kernels.h)
			compute_cov3D(p.scales, p.rotations, (size_t)idx, p.scale_modifier, p.raw_params, c3);
		}
		float tx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
		float ty = V[1] * mx + V[5] * my + V[9] * mz + V[13];
		const float tz0 = V[2] * mx + V[6] * my + V[10] * mz + V[14];
		const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
		const float txtz = tx / tz0, tytz = ty / tz0;
		tx = fminf(limx, fmaxf(-limx, txtz)) * tz0;
		ty = fminf(limy, fmaxf(-limy, tytz)) * tz0;
		const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
		const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
		const float h_x = p.focal_x, h_y = p.focal_y;
		const float J00 = h_x / tz0, J02 = -(h_x * tx) / (tz0 * tz0);
		const float J11 = h_y / tz0, J12 = -(h_y * ty) / (tz0 * tz0);
		// T[c][r], W[c][r] = view[4r + c]  (see preprocess.hip for the glm index algebra)
		const float T00 = V[0] * J00 + V[2] * J02, T01 = V[4] * J00 + V[6] * J02, T02 = V[8] * J00 + V[10] * J02;
		const float T10 = V[1] * J11 + V[2] * J12, T11 = V[5] * J11 + V[6] * J12, T12 = V[9] * J11 + V[10] * J12;
		// Vrk[c][r]
		const float V00 = c3[0], V01 = c3[1], V02 = c3[2], V11 = c3[3], V12 = c3[4], V22 = c3[5];
		const float V10 = V01, V20 = V02, V21 = V12;
		const float A00 = T00 * c3[0] + T01 * c3[1] + T02 * c3[2];
		const float A10 = T00 * c3[1] + T01 * c3[3] + T02 * c3[4];
		const float A20 = T00 * c3[2] + T01 * c3[4] + T02 * c3[5];
		const float A01 = T10 * c3[0] + T11 * c3[1] + T12 * c3[2];
		const float A11 = T10 * c3[1] + T11 * c3[3] + T12 * c3[4];
		const float A21 = T10 * c3[2] + T11 * c3[4] + T12 * c3[5];
		const float a = (A00 * T00 + A10 * T01 + A20 * T02) + 0.3f;
		const float b = A01 * T00 + A11 * T01 + A21 * T02;
		const float c = (A01 * T10 + A11 * T11 + A21 * T12) + 0.3f;
		const float denom = a * c - b * b;
		float dL_da = 0, dL_db = 0, dL_dc = 0;
		const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
		if (denom2inv != 0) {
			dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
			dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
			dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
			dcov[0] = (T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc);
			dcov[3] = (T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc);
			dcov[5] = (T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc);
			dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
			dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
			dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
		}

		const float dL_dT00 = 2 * (T00 * V00 + T01 * V01 + T02 * V02) * dL_da + (T10 * V00 + T11 * V01 + T12 * V02) * dL_db;
		const float dL_dT01 = 2 * (T00 * V10 + T01 * V11 + T02 * V12) * dL_da + (T10 * V10 + T11 * V11 + T12 * V12) * dL_db;
		const float dL_dT02 = 2 * (T00 * V20 + T01 * V21 + T02 * V22) * dL_da + (T10 * V20 + T11 * V21 + T12 * V22) * dL_db;
		const float dL_dT10 = 2 * (T10 * V00 + T11 * V01 + T12 * V02) * dL_dc + (T00 * V00 + T01 * V01 + T02 * V02) * dL_db;
		const float dL_dT11 = 2 * (T10 * V10 + T11 * V11 + T12 * V12) * dL_dc + (T00 * V10 + T01 * V11 + T02 * V12) * dL_db;
		const float dL_dT12 = 2 * (T10 * V20 + T11 * V21 + T12 * V22) * dL_dc + (T00 * V20 + T01 * V21 + T02 * V22) * dL_db;
		// W[c][r] = view[4r + c]
		const float dL_dJ00 = V[0] * dL_dT00 + V[4] * dL_dT01 + V[8] * dL_dT02;
		const float dL_dJ02 = V[2] * dL_dT00 + V[6] * dL_dT01 + V[10] * dL_dT02;
		const float dL_dJ11 = V[1] * dL_dT10 + V[5] * dL_dT11 + V[9] * dL_dT12;
		const float dL_dJ12 = V[2] * dL_dT10 + V[6] * dL_dT11 + V[10] * dL_dT12;
		const float tz = 1.f / tz0;
		const float tz2 = tz * tz;
		const float tz3 = tz2 * tz;
		const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
		const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
		const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
		// transformVec4x3Transpose, auxiliary.h:89-97
		gmx = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
		gmy = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
		gmz = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

		// ------------------------------------------------------------------ preprocessCUDA (bwd), backward.cu:346-396
		{
			const float hw = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
			const float m_w = 1.0f / (hw + 0.0000001f);
			const float g2x = ga0.w, g2y = ga1.x;
			const float mul1 = (Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12]) * m_w * m_w;
			const float mul2 = (Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13]) * m_w * m_w;
			gmx += (Pm[0] * m_w - Pm[3] * mul1) * g2x + (Pm[1] * m_w - Pm[3] * mul2) * g2y;
			gmy += (Pm[4] * m_w - Pm[7] * mul1) * g2x + (Pm[5] * m_w - Pm[7] * mul2) * g2y;
			gmz += (Pm[8] * m_w - Pm[11] * mul1) * g2x + (Pm[9] * m_w - Pm[11] * mul2) * g2y;
		}
	}

	gmx += shx;   // same order as the reference: covariance, projection, then the SH direction term
	gmy += shy;
	gmz += shz;
	if (in_range) {
		p.dL_dmean3D[3 * (size_t)idx + 0] = gmx;
		p.dL_dmean3D[3 * (size_t)idx + 1] = gmy;
		p.dL_dmean3D[3 * (size_t)idx + 2] = gmz;
		if (p.dL_dcov3D) {
#pragma unroll
			for (int i = 0; i < 6; i++) p.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
		}
	}

	// ------------------------------------------------------------------ cov3D backward, backward.cu:278-341
	float g_scale[3] = {0.f, 0.f, 0.f};
	float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
	if (vis && p.scales) {
		float4 q = reinterpret_cast<const float4*>(p.rotations)[idx];
		float qn = 1.0f;
		if (p.raw_params & GSR_RAW_ROTATION) {
			qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
			q.x = q.x / qn;
			q.y = q.y / qn;
			q.z = q.z / qn;
			q.w = q.w / qn;
		}
		const float r = q.x, x = q.y, y = q.z, z = q.w;
		float sx = p.scales[3 * (size_t)idx], sy = p.scales[3 * (size_t)idx + 1], sz = p.scales[3 * (size_t)idx + 2];
		if (p.raw_params & GSR_RAW_SCALING) {
			sx = expf(sx);
			sy = expf(sy);
			sz = expf(sz);
		}
		const float s0 = p.scale_modifier * sx, s1 = p.scale_modifier * sy, s2 = p.scale_modifier * sz;
		// R[c][r]
		const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
		const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
		const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
		// 2*M, M[c][r] = s_r * R[c][r]
		const float N00 = 2.0f * (s0 * R00), N01 = 2.0f * (s1 * R01), N02 = 2.0f * (s2 * R02);
		const float N10 = 2.0f * (s0 * R10), N11 = 2.0f * (s1 * R11), N12 = 2.0f * (s2 * R12);
		const float N20 = 2.0f * (s0 * R20), N21 = 2.0f * (s1 * R21), N22 = 2.0f * (s2 * R22);
		// dL_dSigma[c][r] symmetric
		const float S00 = dcov[0], S01 = 0.5f * dcov[1], S02 = 0.5f * dcov[2];
		const float S11 = dcov[3], S12 = 0.5f * dcov[4], S22 = dcov[5];
		const float S10 = S01, S20 = S02, S21 = S12;
		// dL_dM = (2M) * dL_dSigma : dM[c][r] = N[0][r]*S[c][0] + N[1][r]*S[c][1] + N[2][r]*S[c][2]
#define DM(c_, r_) (N0##r_ * S##c_##0 + N1##r_ * S##c_##1 + N2##r_ * S##c_##2)
		// dL_dMt[c][r] = dL_dM[r][c]
		float D00 = DM(0, 0), D01 = DM(1, 0), D02 = DM(2, 0);
		float D10 = DM(0, 1), D11 = DM(1, 1), D12 = DM(2, 1);
		float D20 = DM(0, 2), D21 = DM(1, 2), D22 = DM(2, 2);
#undef DM
		// Rt[c][r] = R[r][c];  dL_dscale.k = dot(Rt[k], dL_dMt[k])
		// the reference writes dot(Rt[k], dL_dMt[k]) as the scale gradient (backward.cu:318-321); a raw log-scale
		// adds d exp = the activated scale
		const float ds0 = R00 * D00 + R10 * D01 + R20 * D02, ds1 = R01 * D10 + R11 * D11 + R21 * D12,
		            ds2 = R02 * D20 + R12 * D21 + R22 * D22;
		const bool raw_s = (p.raw_params & GSR_RAW_SCALING) != 0;
		g_scale[0] = raw_s ? ds0 * sx : ds0;
		g_scale[1] = raw_s ? ds1 * sy : ds1;
		g_scale[2] = raw_s ? ds2 * sz : ds2;
		D00 *= s0; D01 *= s0; D02 *= s0;
		D10 *= s1; D11 *= s1; D12 *= s1;
		D20 *= s2; D21 *= s2; D22 *= s2;
		dq.x = 2 * z * (D01 - D10) + 2 * y * (D20 - D02) + 2 * x * (D12 - D21);
		dq.y = 2 * y * (D10 + D01) + 2 * z * (D20 + D02) + 2 * r * (D12 - D21) - 4 * x * (D22 + D11);
		dq.z = 2 * x * (D10 + D01) + 2 * r * (D20 - D02) + 2 * z * (D12 + D21) - 4 * y * (D22 + D00);
		dq.w = 2 * r * (D01 - D10) + 2 * x * (D20 + D02) + 2 * y * (D12 + D21) - 4 * z * (D11 + D00);
		// no normalisation Jacobian in the reference kernel (backward.cu:340: autograd's normalize supplies it);
		// for a raw quaternion it is applied here: dL/dr = (g - q (q.g)) / |r|
		if (p.raw_params & GSR_RAW_ROTATION) {
			const float qg = r * dq.x + x * dq.y + y * dq.z + z * dq.w;
			dq.x = (dq.x - r * qg) / qn;
			dq.y = (dq.y - x * qg) / qn;
			dq.z = (dq.z - y * qg) / qn;
			dq.w = (dq.w - z * qg) / qn;
		}
	}
	if (in_range && p.geom.on) {
		geom_adam_row<3>(p.geom.scaling, (size_t)idx, g_scale);
		geom_adam_row4(p.geom.rotation, (size_t)idx, dq);
	} else if (in_range && p.dL_dscale) {
		p.dL_dscale[3 * (size_t)idx + 0] = g_scale[0];
		p.dL_dscale[3 * (size_t)idx + 1] = g_scale[1];
		p.dL_dscale[3 * (size_t)idx + 2] = g_scale[2];
		reinterpret_cast<float4*>(p.dL_drot)[idx] = dq;
	}
}

// The Adam step of the SH rows of the culled Gaussians (zero gradient: m <- b1 m, v <- b2 v, p <- p - step m / (sqrt(v) c + eps)),
// the arithmetic of wave_adam_rows with g = 0.  One thread per float4 of a row (12 per row); rows of visible Gaussians are
// not touched.  HBM-bound streaming (576 B read + 576 B written per culled Gaussian) meant to run on a second stream next to
// the VALU-bound backward blend (gsr_backward).
__device__ __forceinline__ void sh_adam_culled_item(long long t, const int* __restrict__ radii, const RowAdam& a)
{
	const long long row = t / ROW_F4;
	if (radii[row] > 0) return;
	const int col = (int)(t - row * ROW_F4);
	const size_t i = (size_t)t;
	float4 pv = load_stream_f4(reinterpret_cast<const float4*>(a.param) + i);
	float4 mv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg) + i);
	float4 vv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg_sq) + i);
	float* pp = &pv.x; float* mp = &mv.x; float* vp = &vv.x;
	const float ss_first = col == 0 ? a.s.step_size : a.s.step_size_tail;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		const float ss = e < 3 ? ss_first : a.s.step_size_tail;
		const float g = 0.f;
		mp[e] = a.s.b1 * mp[e] + a.s.omb1 * g;
		vp[e] = a.s.b2 * vp[e] + a.s.omb2 * g * g;
		pp[e] -= ss * adam_ratio(mp[e], vp[e], a.s.inv_sqrt_bc2, a.s.eps);
	}
	store_stream_f4(reinterpret_cast<float4*>(a.param) + i, pv);
	store_stream_f4(reinterpret_cast<float4*>(a.exp_avg) + i, mv);
	store_stream_f4(reinterpret_cast<float4*>(a.exp_avg_sq) + i, vv);
}

__global__ void __launch_bounds__(256)
sh_adam_culled_kernel(int P, const int* __restrict__ radii, const RowAdam a)
{
	// grid-stride over the float4s of the tensor: the grid is kept SMALL on purpose (launch_sh_adam_culled) -- this kernel
	// shares the machine with the backward blend and must not take its wave slots
	const long long items = (long long)P * ROW_F4, stride = (long long)gridDim.x * 256;
	for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < items; t += stride)
		sh_adam_culled_item(t, radii, a);
}

int launch_sh_adam_culled(int P, const int* radii, const RowAdam& adam, hipStream_t stream, int max_blocks)
{
	if (P <= 0) return GSR_OK;
	if ((reinterpret_cast<uintptr_t>(adam.param) | reinterpret_cast<uintptr_t>(adam.exp_avg) | reinterpret_cast<uintptr_t>(adam.exp_avg_sq)) & 15)
		return GSR_ERR_UNSUPPORTED;
	const long long items = (long long)P * ROW_F4;
	// max_blocks (gsr_sh_adam.side_blocks; default 256) -- measured at C3: 256 blocks 1.927 ms / step, 1024: 1.954, 2048: 2.075, no side stream: 2.004
	long long blocks = (items + 255) / 256;
	if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
	GSR_LAUNCH(sh_adam_culled_kernel, (int)blocks, 256, stream, P, radii, adam);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// Lazy mode (shrows.h): one workgroup per row block of LAZY_BLOCK_ROWS rows, one thread per 16-byte vector of the block
// (64 rows x 12 vectors = 768 threads: the `window` dependent steps per element are latency-bound, so the parallelism is spent
// across elements).  The blocks of this step's slice (all blocks: flush) bring their rows with radii <= 0 (radii == null: every
// row that is behind -- the flush, and the slice of the data-parallel program, which runs BEHIND the kernel that stepped the
// lit rows) up to a.step -- the zero-gradient steps (row_step, a.step] in one read-modify-write of the row.  The workgroup owns its
// block: row_step is read by all threads before the barrier, written behind it.
constexpr int LAZY_THREADS = LAZY_BLOCK_ROWS * ROW_F4;   // 768
// mode 0: this step's slice (row blocks b with b % window == step % window), rows with radii <= 0, up to a.step (the fused
//         backward: the visible rows step in sh_bwd_rows_kernel at the same time);
// mode 1: every block, every row that is behind, up to a.step (gsr_sh_adam_flush; radii == null);
// mode 2: this step's slice, every row that is behind, up to a.step (gsr_sh_adam_lazy_slice: BEHIND the kernel that stepped
//         the lit rows of a data-parallel step);
// mode 3: the slice of period window - 1, every row that is behind, up to a.step - 1 only (the data-parallel step's catch-up
//         run AHEAD of the exchange, next to the backward blend: what happens at a.step is not known yet -- a row some view
//         lights takes a real step, the others none -- so a row visited here lags by at most window - 1 steps afterwards).
__global__ void __launch_bounds__(LAZY_THREADS)
sh_adam_lazy_kernel(int P, const int* __restrict__ radii, const LazyAdam a, int mode)
{
	const int period = mode == 3 ? a.window - 1 : a.window;
	const long long b = mode == 1 ? (long long)blockIdx.x : (long long)(a.step % period) + (long long)blockIdx.x * period;
	const size_t row0 = (size_t)b * LAZY_BLOCK_ROWS;
	const int item = (int)threadIdx.x;
	const int rr = item / ROW_F4, col = item - rr * ROW_F4;
	const size_t row = row0 + (size_t)rr;
	const bool mine = row < (size_t)P && (radii == nullptr || radii[row] <= 0);
	const int k_lo = mode == 3 ? 1 : 0;
	bool moved = false;
	if (mine) {
		int k_hi = a.step - 1 - a.row_step[row];
		k_hi = k_hi >= a.window ? a.window - 1 : k_hi;
		if (k_hi >= k_lo) {
			moved = true;
			const size_t i = row * ROW_F4 + col;
			float4 pv = load_stream_f4(reinterpret_cast<const float4*>(a.param) + i);
			float4 mv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg) + i);
			float4 vv = load_stream_f4(reinterpret_cast<const float4*>(a.exp_avg_sq) + i);
			lazy_zero_grad_steps(a.t, k_hi, k_lo, col, pv, mv, vv);
			store_stream_f4(reinterpret_cast<float4*>(a.param) + i, pv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg) + i, mv);
			store_stream_f4(reinterpret_cast<float4*>(a.exp_avg_sq) + i, vv);
		}
	}
	__syncthreads();
	if (mode == 3) {
		if (moved && col == 0) a.row_step[row] = a.step - 1;
	} else if (mine && col == 0) {
		a.row_step[row] = a.step;
	}
}

int launch_sh_adam_lazy(int P, const int* radii, const LazyAdam& a, hipStream_t stream, int mode)
{
	if (P <= 0) return GSR_OK;
	if (!a.row_step || a.window < 2 || a.window > LAZY_WINDOW_MAX || a.step < 1) return GSR_ERR_INVALID_ARG;
	if (mode == 3 && a.window < 3) return GSR_ERR_INVALID_ARG;
	if ((radii != nullptr) != (mode == 0)) return GSR_ERR_INVALID_ARG;
	if ((reinterpret_cast<uintptr_t>(a.param) | reinterpret_cast<uintptr_t>(a.exp_avg) | reinterpret_cast<uintptr_t>(a.exp_avg_sq)) & 15)
		return GSR_ERR_UNSUPPORTED;
	const int nb = div_up(P, LAZY_BLOCK_ROWS);
	// slice: the row blocks b with b % period == step % period
	const int period = mode == 3 ? a.window - 1 : a.window;
	const int phase = a.step % period;
	const int blocks = mode == 1 ? nb : (nb > phase ? div_up(nb - phase, period) : 0);
	if (blocks > 0) GSR_LAUNCH(sh_adam_lazy_kernel, blocks, LAZY_THREADS, stream, P, radii, a, mode);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// The runs of more than LONG_RUN instance slots (partials.h): one wave per run, a fixed grid that strides over the list the
// forward pass left (its length lives on the device).
constexpr int LRS_BLOCKS = 1024;   // (one listed run per wave at C3: 512 -> 1 024 workgroups: 18.6 -> 14.1 us; more: equal)
__global__ void __launch_bounds__(256)
long_run_sums_kernel(const PreprocessBwdParams p)
{
	// wave -> (sub-list, position): consecutive waves take different sub-lists, waves/LONG_LISTS of them share one
	const uint32_t wave = (uint32_t)blockIdx.x * 4u + (uint32_t)wave_id(), waves = (uint32_t)gridDim.x * 4u;
	const uint32_t list = wave % (uint32_t)LONG_LISTS;
	const uint32_t n = p.long_counts[list * LONG_COUNT_STRIDE];
	for (uint32_t e = wave / (uint32_t)LONG_LISTS; e < n; e += waves / (uint32_t)LONG_LISTS) {
		const uint32_t g = p.long_runs[(size_t)list * p.long_capacity + e];
		const uint32_t cnt = p.tiles_touched[g];
		const uint32_t first = __float_as_uint(p.rec[3 * (size_t)g + 2].w);
		wave_sum_long_run(first, cnt, p.partials, p.touched);
	}
}

int launch_preprocess_bwd(const PreprocessBwdParams& p, hipStream_t stream)
{
	if (p.partials) GSR_LAUNCH(long_run_sums_kernel, LRS_BLOCKS, 256, stream, p);
	const bool factored = p.dL_dcolor_view != nullptr;
	const bool adam = p.adam_exp_avg != nullptr;
	const bool rows_ok = sh_rows_path(p.shs, p.M, p.D, factored, adam, p.dL_dsh);   // (includes 0 <= D <= 3)
	if (adam && (!rows_ok || factored || ((reinterpret_cast<uintptr_t>(p.adam_exp_avg) | reinterpret_cast<uintptr_t>(p.adam_exp_avg_sq)) & 15)))
		return GSR_ERR_UNSUPPORTED;   // the fused step exists for aligned [P,16,3] rows only
	if (p.geom.on && !(rows_ok && p.D >= 0 && p.D <= 3 && p.scales && p.rotations && p.dL_dmean3D))
		return GSR_ERR_UNSUPPORTED;   // the fused geometry step lives in the two-kernel path of the reference's SH layout
	const int grid = div_up(p.P, PRB_THREADS);
	if (rows_ok && p.D >= 0 && p.D <= 3) {
		if (p.packed_msg) GSR_LAUNCH((preprocess_bwd_kernel<true, true>), grid, PRB_THREADS, stream, p);
		else GSR_LAUNCH(preprocess_bwd_kernel<true>, grid, PRB_THREADS, stream, p);
		GSR_CHECK_LAUNCH();
		if (p.notify_stream && p.notify_event) {   // dL_dcolor_view is complete: whoever gathers it need not wait for the SH kernel
			GSR_HIP(hipEventRecord((hipEvent_t)p.notify_event, stream));
			GSR_HIP(hipStreamWaitEvent((hipStream_t)p.notify_stream, (hipEvent_t)p.notify_event, 0));
		}
		const int g = div_up(p.P, SHB_THREADS);
#define GSR_SHB(DEG)                                                                  \
	do {                                                                              \
		if (factored)                                                                 \
			GSR_LAUNCH((sh_bwd_rows_kernel<DEG, 1>), g, SHB_THREADS, stream, p);      \
		else if (adam)                                                                \
			GSR_LAUNCH((sh_bwd_rows_kernel<DEG, 2>), g, SHB_THREADS, stream, p);      \
		else                                                                          \
			GSR_LAUNCH((sh_bwd_rows_kernel<DEG, 0>), g, SHB_THREADS, stream, p);      \
	} while (0)
		if (p.D == 3)
			GSR_SHB(3);
		else if (p.D == 2)
			GSR_SHB(2);
		else if (p.D == 1)
			GSR_SHB(1);
		else
			GSR_SHB(0);
#undef GSR_SHB
	} else {
		if (p.packed_msg) GSR_LAUNCH((preprocess_bwd_kernel<false, true>), grid, PRB_THREADS, stream, p);
		else GSR_LAUNCH(preprocess_bwd_kernel<false>, grid, PRB_THREADS, stream, p);
		if (p.notify_stream && p.notify_event) {
			GSR_HIP(hipEventRecord((hipEvent_t)p.notify_event, stream));
			GSR_HIP(hipStreamWaitEvent((hipStream_t)p.notify_stream, (hipEvent_t)p.notify_event, 0));
		}
	}
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Keyframe-batch data parallelism.  The SH gradient of ONE view is rank one per Gaussian:
// dL_dsh[k][ch] = basis_k(dir) * dRGB[ch] (backward.cu:41-127), and dir = normalize(mean - campos) is known to every
// rank.  So the ranks exchange the 3 floats dRGB per view (all-gather) instead of reducing the 48-float rows, and
// each rebuilds   dL_dsh[i][k][ch] = scale * sum_v basis_k(dir_v(i)) * dRGB_v[i][ch]   here: 12 n_views B read and
// 192 B written per Gaussian, against 2 * 192 B sent per Gaussian by a ring all-reduce of the rows.
// ADAM: the rows do not leave as a gradient -- this step's Adam update of the SH tensor is applied from LDS
// (gsr_sh_adam_from_views; wave_adam_rows, shrows.h).
// LAZY (gsr_sh_adam_from_views with sh_adam->lazy): a row whose colour gradient is zero in EVERY gathered view takes a
// zero-gradient step -- exactly the case the lazy rows of the single-GPU program defer (shrows.h) -- so it is left alone here
// (row_step keeps counting what it has taken); a row with a gradient first takes the zero-gradient steps it is behind (the
// forward pass only caught up the rows THIS rank's view sees; another rank's view may light a row this rank culled), then this
// step, and row_step[i] = step.  The dense 1152 B per Gaussian become 1152 B per Gaussian SOME view of the batch sees.
// PACKED: the views arrive as the messages of gsr_pack_color_view (include/gsr.h) -- a bit per Gaussian says whether the view sees
// it, its row sits at (seen rows in front of the 64-row group) + (seen rows in front of it inside the group); the camera centre
// rides in the header.  Same rows, same order of the views: the same bits as the dense form.
template <int DEG, bool ADAM, bool LAZY, bool PACKED>
__global__ void __launch_bounds__(SHB_THREADS) GSR_WAVES_PER_EU(4, 8)
sh_grad_from_views_kernel(int P, int n_views, const float* __restrict__ means3D, const float* __restrict__ campos,
                          long long campos_stride, const float* __restrict__ views, long long view_stride, float scale,
                          float* __restrict__ dL_dsh, const RowAdam adam, const LazyAdam lz, const PackedViews pk,
                          int pk_prefix_words, int pk_mask_words)
{
	__shared__ float4 s_rows[SHB_THREADS / 64][STAGE_ROWS][ROW_F4_PAD];
	__shared__ uint32_t s_lag[SHB_THREADS / 64][STAGE_ROWS];
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int w = wave_id(), l = lane_id();
	const size_t wave_first = (size_t)(blockIdx.x * blockDim.x) + (size_t)w * 64;
	const bool in_range = idx < P;
	constexpr int ncoef = (DEG + 1) * (DEG + 1);
	float acc[3 * ncoef];
#pragma unroll
	for (int j = 0; j < 3 * ncoef; j++) acc[j] = 0.f;
	float mx = 0.f, my = 0.f, mz = 0.f;
	if (in_range) {
		mx = means3D[3 * (size_t)idx];
		my = means3D[3 * (size_t)idx + 1];
		mz = means3D[3 * (size_t)idx + 2];
	}
	bool any = false;   // some view holds a colour gradient for this Gaussian
#pragma unroll 1
	for (int v = 0; v < n_views; v++) {
		float r = 0.f, g = 0.f, b = 0.f;
		const uint32_t* msg = nullptr;
		if (PACKED) {
			msg = pk.msgs + (size_t)v * (size_t)pk.stride;
			// (wave-uniform header words.  A message written for another P is not decoded at all, and no row beyond the rows the
			// message HOLDS is read -- a sender whose capacity was too small has set msg[3] and dropped them; gsr_check_packed_views
			// is the loud form of the same test)
			// (msg[2] is the capacity of the SENDER's buffer -- gsr_backward writes its message with room for every row -- which may
			// exceed the rows that travelled: the stride between two gathered messages says how many rows one of them has room for)
			const long long room = (pk.stride - (long long)(PACK_HEADER + pk_prefix_words + pk_mask_words + 4)) / 3;
			const uint32_t travelled = room > 0 ? (uint32_t)(room < 0x7FFFFFFFll ? room : 0x7FFFFFFFll) : 0u;
			uint32_t held = msg[1] == (uint32_t)P ? (msg[0] < msg[2] ? msg[0] : msg[2]) : 0u;
			held = held < travelled ? held : travelled;
			if (in_range) {
				const uint32_t* prefix = msg + PACK_HEADER;
				const unsigned long long mw = reinterpret_cast<const unsigned long long*>(prefix + pk_prefix_words)[idx >> 6];
				if ((mw >> (idx & 63)) & 1ull) {
					const uint32_t rank = prefix[idx >> 6] + (uint32_t)__popcll(mw & ((1ull << (idx & 63)) - 1ull));
					if (rank < held) {
						const float* c = reinterpret_cast<const float*>(prefix + pk_prefix_words + pk_mask_words) + 3 * (size_t)rank;
						r = c[0]; g = c[1]; b = c[2];
					}
				}
			}
		} else if (in_range) {
			const float* c = views + (size_t)v * (size_t)view_stride + (size_t)idx * 3;
			r = c[0]; g = c[1]; b = c[2];
		}
		// some view SEES this Gaussian (a visible one with an all-zero gradient carries -0.0f: sh_bwd_rows_kernel)
		any = any || (__float_as_uint(r) | __float_as_uint(g) | __float_as_uint(b)) != 0u;
		// culled in this view (or every channel clamped): nothing to add, and most Gaussians are outside most views
		if (r != 0.f || g != 0.f || b != 0.f) {
			const float* cp = PACKED ? reinterpret_cast<const float*>(msg + 4) : campos + (size_t)v * (size_t)campos_stride;
			const float ox = mx - cp[0], oy = my - cp[1], oz = mz - cp[2];
			const float len = sqrtf(ox * ox + oy * oy + oz * oz);   // forward.cu:27-28
			const ShDir d = sh_dir(ox / len, oy / len, oz / len);
#pragma unroll
			for (int k = 0; k < ncoef; k++) {
				const float bk = sh_basis(k, d);
				acc[3 * k + 0] += bk * r;
				acc[3 * k + 1] += bk * g;
				acc[3 * k + 2] += bk * b;
			}
		}
	}
	int lag = 0;
	if (LAZY && any) {
		lag = lz.step - 1 - lz.row_step[idx];
		lag = lag < 0 ? 0 : (lag >= lz.window ? lz.window - 1 : lag);
	}
	// rows leave through LDS as contiguous 6 KiB bursts, half a wave at a time (shrows.h)
#pragma unroll 1
	for (int h = 0; h < 64 / STAGE_ROWS; h++) {
		const size_t half_first = wave_first + (size_t)(h * STAGE_ROWS);
		const long long left = (long long)P - (long long)half_first;
		if (left <= 0) break;   // wave-uniform
		const bool mine = (l / STAGE_ROWS) == h;
		const unsigned long long m = LAZY ? wave_ballot(any && mine) : ~0ull;
		if (mine && (!LAZY || any)) {
			float4* row = s_rows[w][l % STAGE_ROWS];
#pragma unroll
			for (int i = 0; i < ROW_F4; i++) {
				float o[4];
#pragma unroll
				for (int c = 0; c < 4; c++) o[c] = (4 * i + c < 3 * ncoef) ? acc[(4 * i + c < 3 * ncoef) ? 4 * i + c : 0] * scale : 0.f;
				row[i] = make_float4(o[0], o[1], o[2], o[3]);
			}
			if (LAZY) s_lag[w][l % STAGE_ROWS] = (uint32_t)lag;
		}
		if (ADAM && LAZY) {
			if (m) wave_adam_rows<true>(adam, half_first, (int)(left > STAGE_ROWS ? STAGE_ROWS : left), s_rows[w],
			                            (uint32_t)(m >> (h * STAGE_ROWS)), &lz.t, s_lag[w]);
		} else if (ADAM)
			wave_adam_rows(adam, half_first, (int)(left > STAGE_ROWS ? STAGE_ROWS : left), s_rows[w]);
		else
			wave_store_rows(reinterpret_cast<float4*>(dL_dsh), half_first, (int)(left > STAGE_ROWS ? STAGE_ROWS : left), s_rows[w]);
	}
	if (LAZY && any) lz.row_step[idx] = lz.step;   // this row has taken the step
}

// any other row length / alignment: one thread per Gaussian, scalar stores
__global__ void __launch_bounds__(128)
sh_grad_from_views_generic_kernel(int P, int D, int M, int n_views, const float* __restrict__ means3D,
                                  const float* __restrict__ campos, long long campos_stride,
                                  const float* __restrict__ views, long long view_stride, float scale,
                                  float* __restrict__ dL_dsh)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (idx >= P) return;
	const int ncoef = (D + 1) * (D + 1);
	float acc[48];
#pragma unroll
	for (int j = 0; j < 48; j++) acc[j] = 0.f;
	const float mx = means3D[3 * (size_t)idx], my = means3D[3 * (size_t)idx + 1], mz = means3D[3 * (size_t)idx + 2];
	for (int v = 0; v < n_views; v++) {
		const float* c = views + (size_t)v * (size_t)view_stride + (size_t)idx * 3;
		const float r = c[0], g = c[1], b = c[2];
		if (r == 0.f && g == 0.f && b == 0.f) continue;
		const float* cp = campos + (size_t)v * (size_t)campos_stride;
			const float ox = mx - cp[0], oy = my - cp[1], oz = mz - cp[2];
		const float len = sqrtf(ox * ox + oy * oy + oz * oz);
		const ShDir d = sh_dir(ox / len, oy / len, oz / len);
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (k < ncoef) {
				const float bk = sh_basis(k, d);
				acc[3 * k + 0] += bk * r;
				acc[3 * k + 1] += bk * g;
				acc[3 * k + 2] += bk * b;
			}
		}
	}
	float* out = dL_dsh + (size_t)idx * 3 * M;
#pragma unroll
	for (int j = 0; j < 48; j++)
		if (j < 3 * ncoef) out[j] = acc[j] * scale;
	for (int j = 3 * ncoef; j < 3 * M; j++) out[j] = 0.f;
}

int launch_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                              long long campos_stride, const float* views, long long view_stride, float scale, float* dL_dsh,
                              const RowAdam* adam, hipStream_t stream, const LazyAdam* lazy, PackedViews packed)
{
	if (P == 0) return GSR_OK;
	const int pkp = (int)pack_prefix_words(P), pkm = (int)pack_mask_words(P);
	if (campos_stride == 0) campos_stride = 3;
	if (view_stride == 0) view_stride = 3ll * P;
	float* rows = adam ? adam->param : dL_dsh;
	const bool rows_ok = (3 * M == ROW_F4 * 4) && ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
	if (adam && (!rows_ok || ((reinterpret_cast<uintptr_t>(adam->exp_avg) | reinterpret_cast<uintptr_t>(adam->exp_avg_sq)) & 15)))
		return GSR_ERR_UNSUPPORTED;   // the fused step exists for aligned [P,16,3] rows only
	if (lazy && !adam) return GSR_ERR_INVALID_ARG;
	if (packed.msgs && !rows_ok) return GSR_ERR_UNSUPPORTED;   // the packed form exists for aligned [P,16,3] rows only
	if (rows_ok) {
		const int g = div_up(P, SHB_THREADS);
		const RowAdam ra = adam ? *adam : RowAdam{};
		const LazyAdam lz = lazy ? *lazy : LazyAdam{};
#define GSR_SHV_LAUNCH(DEG, A, L, K)                                                                                          \
	GSR_LAUNCH((sh_grad_from_views_kernel<DEG, A, L, K>), g, SHB_THREADS, stream, P, n_views, means3D, campos, campos_stride,   \
	           views, view_stride, scale, dL_dsh, ra, lz, packed, pkp, pkm)
#define GSR_SHV(DEG)                                                                                                          \
	do {                                                                                                                      \
		if (packed.msgs) {                                                                                                    \
			if (adam && lazy) GSR_SHV_LAUNCH(DEG, true, true, true);                                                          \
			else if (adam) GSR_SHV_LAUNCH(DEG, true, false, true);                                                            \
			else GSR_SHV_LAUNCH(DEG, false, false, true);                                                                     \
		} else {                                                                                                              \
			if (adam && lazy) GSR_SHV_LAUNCH(DEG, true, true, false);                                                         \
			else if (adam) GSR_SHV_LAUNCH(DEG, true, false, false);                                                           \
			else GSR_SHV_LAUNCH(DEG, false, false, false);                                                                    \
		}                                                                                                                     \
	} while (0)
		if (D == 3)
			GSR_SHV(3);
		else if (D == 2)
			GSR_SHV(2);
		else if (D == 1)
			GSR_SHV(1);
		else
			GSR_SHV(0);
#undef GSR_SHV
#undef GSR_SHV_LAUNCH
	} else {
		GSR_LAUNCH(sh_grad_from_views_generic_kernel, div_up(P, 128), 128, stream, P, D, M, n_views, means3D, campos,
		           campos_stride, views, view_stride, scale, dL_dsh);
	}
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// gsr_pack_color_view (include/gsr.h): one view's [P,3] colour gradient as a message of its SEEN rows.  A row is seen when any
// of its three words is non-zero (a lit row, or the -0.0f marker of a visible Gaussian whose gradient is all zero); the order
// of the packed rows is the index order, so the decoder needs only the count in front of each 64-row group (an exclusive scan
// of the groups' popcounts, launch_scan_u32) and the 64-bit mask of the group.
__global__ void __launch_bounds__(256)
pack_mask_kernel(int P, const float* __restrict__ view, uint32_t* __restrict__ counts, uint32_t* __restrict__ mask)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	bool seen = false;
	if (idx < P) {
		const float* c = view + 3 * (size_t)idx;
		seen = (__float_as_uint(c[0]) | __float_as_uint(c[1]) | __float_as_uint(c[2])) != 0u;
	}
	const unsigned long long m = wave_ballot(seen);
	const int group = idx >> 6;   // wave-uniform: blocks of 256 threads start at multiples of 64
	if (lane_id() == 0 && (size_t)group < (((size_t)P + 63) >> 6)) {
		counts[group] = (uint32_t)__popcll(m);
		mask[2 * (size_t)group] = (uint32_t)m;
		mask[2 * (size_t)group + 1] = (uint32_t)(m >> 32);
	}
}

// the same two sections from the forward pass's radii (gsr_pack_view_plan): a Gaussian is seen when its radius is positive --
// exactly the rows preprocess_bwd_kernel leaves non-zero in the view (a lit row or the -0.0f marker)
__global__ void __launch_bounds__(256)
pack_plan_kernel(int P, const int* __restrict__ radii, uint32_t* __restrict__ counts, uint32_t* __restrict__ mask)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const bool seen = idx < P && radii[idx] > 0;
	const unsigned long long m = wave_ballot(seen);
	const int group = idx >> 6;
	if (lane_id() == 0 && (size_t)group < (((size_t)P + 63) >> 6)) {
		counts[group] = (uint32_t)__popcll(m);
		mask[2 * (size_t)group] = (uint32_t)m;
		mask[2 * (size_t)group + 1] = (uint32_t)(m >> 32);
	}
}

__global__ void __launch_bounds__(256)
pack_rows_kernel(int P, const float* __restrict__ view, const float* __restrict__ campos, int capacity, uint32_t* __restrict__ msg,
                 int prefix_words, int mask_words)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const uint32_t* prefix = msg + PACK_HEADER;
	float* rows = reinterpret_cast<float*>(msg + PACK_HEADER + prefix_words + mask_words);
	float r = 0.f, g = 0.f, b = 0.f;
	bool seen = false;
	if (idx < P) {
		const float* c = view + 3 * (size_t)idx;
		r = c[0]; g = c[1]; b = c[2];
		seen = (__float_as_uint(r) | __float_as_uint(g) | __float_as_uint(b)) != 0u;
	}
	const unsigned long long m = wave_ballot(seen);
	const int group = idx >> 6;
	const size_t groups = ((size_t)P + 63) >> 6;   // (pack_groups: a host helper)
	const bool group_exists = (size_t)group < groups;
	const uint32_t first = group_exists ? prefix[group] : 0u;
	if (seen) {
		const uint32_t rank = first + (uint32_t)__popcll(m & lanemask_lt());
		if (rank < (uint32_t)capacity) {
			rows[3 * (size_t)rank] = r;
			rows[3 * (size_t)rank + 1] = g;
			rows[3 * (size_t)rank + 2] = b;
		}
	}
	if (group_exists && (size_t)group == groups - 1 && lane_id() == 0) {   // the last group knows the total
		const uint32_t K = first + (uint32_t)__popcll(m);
		msg[0] = K;
		msg[1] = (uint32_t)P;
		msg[2] = (uint32_t)capacity;
		msg[3] = K > (uint32_t)capacity ? 1u : 0u;
		msg[4] = __float_as_uint(campos[0]);
		msg[5] = __float_as_uint(campos[1]);
		msg[6] = __float_as_uint(campos[2]);
		msg[7] = 0u;
	}
}

int launch_pack_color_view(int P, const float* view, const float* campos, int capacity, uint32_t* msg, uint32_t* scratch, hipStream_t stream)
{
	if (P <= 0) return GSR_OK;
	const int groups = (int)pack_groups(P), pw = (int)pack_prefix_words(P), mw = (int)pack_mask_words(P);
	uint32_t* prefix = msg + PACK_HEADER;
	GSR_LAUNCH(pack_mask_kernel, div_up(P, 256), 256, stream, P, view, prefix, prefix + pw);
	// the groups' counts -> the seen rows in front of each group, in place
	int st = launch_scan_u32(prefix, nullptr, prefix, groups, false, scratch, stream);
	if (st != GSR_OK) return st;
	GSR_LAUNCH(pack_rows_kernel, div_up(P, 256), 256, stream, P, view, campos, capacity, msg, pw, mw);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_pack_view_plan(int P, const int* radii, uint32_t* msg, uint32_t* scratch, hipStream_t stream)
{
	if (P <= 0) return GSR_OK;
	const int groups = (int)pack_groups(P), pw = (int)pack_prefix_words(P);
	uint32_t* prefix = msg + PACK_HEADER;
	GSR_LAUNCH(pack_plan_kernel, div_up(P, 256), 256, stream, P, radii, prefix, prefix + pw);
	int st = launch_scan_u32(prefix, nullptr, prefix, groups, false, scratch, stream);
	if (st != GSR_OK) return st;
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
