// blend_bwd.hip -- backward of the alpha blend: per-pixel loss gradients -> per-Gaussian
// gradients of colour, 2D mean, conic and opacity.
//
// Per-pixel semantics are renderCUDA's backward (cuda_rasterizer/backward.cu:399-557): walk
// the tile list back to front starting at each pixel's last contributor, recompute alpha,
// un-blend T, and accumulate the nine partial derivatives.
//
// The reference issues nine global float atomics per contributing (pixel, Gaussian) pair
// (backward.cu:523-554).  Here:
//   * the same per-quad rejection masks as the forward pass remove whole-wave work;
//   * list entries behind the deepest last-contributor of the tile are never staged, entries
//     behind the wave's deepest one are never visited;
//   * the nine terms are summed across the 64 pixels of a quad with DPP row operations (no
//     LDS, no atomics), the four quad sums meet in LDS (ds_add_f32), and one thread per
//     staged entry flushes them: 9 global atomics per (tile, Gaussian) instead of per
//     (pixel, Gaussian) -- up to 256x fewer, and the order inside a tile is fixed.
#include "blend.h"
#include "kernels.h"

namespace gsr {


__global__ void __launch_bounds__(256)
blend_bwd_kernel(const BlendBwdParams p)
{
	__shared__ float4 s_q0[256];
	__shared__ float4 s_q1[256];
	__shared__ float s_b[256];
	__shared__ uint32_t s_gid[256];
	__shared__ float s_acc[9][256];
	__shared__ uint32_t s_touched[256];
	__shared__ unsigned long long s_mask[4][4];  // [quad][loader wave]
	__shared__ uint32_t s_wmax[4];

	const int tile = xcd_tile((int)blockIdx.x, p.tiles);
	if (tile >= p.tiles) return;
	const int tile_x = tile % p.grid_x, tile_y = tile / p.grid_x;
	const int w = wave_id(), l = lane_id(), tid = (int)threadIdx.x;
	int px, py;
	quad_pixel(tile_x, tile_y, px, py);
	const bool inside = px < p.W && py < p.H;
	const float pxf = (float)px, pyf = (float)py;
	const uint2 range = p.ranges[tile];
	const size_t pix = (size_t)py * p.W + px;
	const size_t plane = (size_t)p.H * p.W;

	const float T_final = inside ? p.final_T[pix] : 0.f;
	float T = T_final;
	const uint32_t last_contributor = inside ? p.n_contrib[pix] : 0u;
	float dpr = 0.f, dpg = 0.f, dpb = 0.f;
	if (inside) {
		dpr = p.dL_dpix[pix];
		dpg = p.dL_dpix[plane + pix];
		dpb = p.dL_dpix[2 * plane + pix];
	}
	const float bg_dot_dpixel = p.bg[0] * dpr + p.bg[1] * dpg + p.bg[2] * dpb;
	float acr = 0.f, acg = 0.f, acb = 0.f;      // accum_rec
	float last_alpha = 0.f, lcr = 0.f, lcg = 0.f, lcb = 0.f;
	const float ddelx_dx = 0.5f * (float)p.W, ddely_dy = 0.5f * (float)p.H;

	// deepest contributor of the wave / of the tile
	const uint32_t wmax = wave_max_u32(last_contributor);
	if (l == 0) s_wmax[w] = wmax;
#pragma unroll
	for (int c = 0; c < 9; c++) s_acc[c][tid] = 0.f;
	s_touched[tid] = 0u;
	__syncthreads();
	const uint32_t bmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
	const int nbatches = (int)((bmax + 255u) >> 8);

	for (int b = nbatches - 1; b >= 0; b--) {
		// ---- stage entries [b*256, b*256+256) below bmax
		const uint32_t e = (uint32_t)(b << 8) + (uint32_t)tid;
		uint32_t keep = 0;
		if (e < bmax) {
			const uint32_t gid = p.point_list[range.x + e];
			const float4 q0 = p.rec[3 * (size_t)gid + 0];
			const float4 q1 = p.rec[3 * (size_t)gid + 1];
			const float4 q2 = p.rec[3 * (size_t)gid + 2];
			s_q0[tid] = q0;
			s_q1[tid] = q1;
			s_b[tid] = q2.x;
			s_gid[tid] = gid;
			keep = quad_keep_bits(q0, q1, (float)(tile_x * TILE), (float)(tile_y * TILE));
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const unsigned long long m = wave_ballot((keep >> q) & 1u);
			if (l == 0) s_mask[q][w] = m;
		}
		__syncthreads();

		// ---- consume back to front
		if ((uint32_t)(b << 8) < wmax) {
			for (int lw = 3; lw >= 0; lw--) {
				unsigned long long m = wave_uniform_u64(s_mask[w][lw]);
				// drop entries at or behind the wave's deepest contributor
				const long long first = (long long)(b << 8) + (lw << 6);
				const long long lim = (long long)wmax - first;  // entries with bit >= lim are not needed
				if (lim <= 0) continue;
				if (lim < 64) m &= (1ull << lim) - 1ull;
				while (m) {
					const int bit = 63 - __clzll((long long)m);
					m &= ~(1ull << bit);
					const int jj = (lw << 6) + bit;
					const uint32_t pos = (uint32_t)(b << 8) + (uint32_t)jj;
					const float4 q0 = s_q0[jj];
					const float4 q1 = s_q1[jj];
					const float cb = s_b[jj];
					float v[9];
#pragma unroll
					for (int c = 0; c < 9; c++) v[c] = 0.f;
					bool contributes = false;
					if (pos < last_contributor) {
						const float dx = q0.x - pxf, dy = q0.y - pyf;
						const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
						if (!(power > 0.0f)) {
							const float G = __expf(power);
							const float alpha = fminf(0.99f, q1.y * G);
							if (!(alpha < 1.0f / 255.0f)) {
								contributes = true;
								T = T / (1.f - alpha);
								const float dchannel_dcolor = alpha * T;
								float dL_dalpha;
								acr = last_alpha * lcr + (1.f - last_alpha) * acr;
								lcr = q1.z;
								dL_dalpha = (q1.z - acr) * dpr;
								acg = last_alpha * lcg + (1.f - last_alpha) * acg;
								lcg = q1.w;
								dL_dalpha += (q1.w - acg) * dpg;
								acb = last_alpha * lcb + (1.f - last_alpha) * acb;
								lcb = cb;
								dL_dalpha += (cb - acb) * dpb;
								v[0] = dchannel_dcolor * dpr;
								v[1] = dchannel_dcolor * dpg;
								v[2] = dchannel_dcolor * dpb;
								dL_dalpha *= T;
								last_alpha = alpha;
								dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
								const float dL_dG = q1.y * dL_dalpha;
								const float gdx = G * dx, gdy = G * dy;
								const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
								const float dG_ddely = -gdy * q1.x - gdx * q0.w;
								v[3] = dL_dG * dG_ddelx * ddelx_dx;
								v[4] = dL_dG * dG_ddely * ddely_dy;
								v[5] = -0.5f * gdx * dx * dL_dG;
								v[6] = -0.5f * gdx * dy * dL_dG;
								v[7] = -0.5f * gdy * dy * dL_dG;
								v[8] = G * dL_dalpha;
							}
						}
					}
					if (wave_ballot(contributes) == 0ull) continue;  // wave-uniform
					wave_reduce9_f32(v);
					if (l == 63) {
#pragma unroll
						for (int c = 0; c < 9; c++) atomicAdd(&s_acc[c][jj], v[c]);
						s_touched[jj] = 1u;
					}
				}
			}
		}
		__syncthreads();

		// ---- flush: thread t owns staged entry t
		if (s_touched[tid]) {
			const uint32_t gid = s_gid[tid];
			atomicAdd(&p.dL_dcolor[3 * (size_t)gid + 0], s_acc[0][tid]);
			atomicAdd(&p.dL_dcolor[3 * (size_t)gid + 1], s_acc[1][tid]);
			atomicAdd(&p.dL_dcolor[3 * (size_t)gid + 2], s_acc[2][tid]);
			atomicAdd(&p.dL_dmean2D[3 * (size_t)gid + 0], s_acc[3][tid]);
			atomicAdd(&p.dL_dmean2D[3 * (size_t)gid + 1], s_acc[4][tid]);
			atomicAdd(&p.dL_dconic[4 * (size_t)gid + 0], s_acc[5][tid]);
			atomicAdd(&p.dL_dconic[4 * (size_t)gid + 1], s_acc[6][tid]);
			atomicAdd(&p.dL_dconic[4 * (size_t)gid + 3], s_acc[7][tid]);
			atomicAdd(&p.dL_dopacity[gid], s_acc[8][tid]);
#pragma unroll
			for (int c = 0; c < 9; c++) s_acc[c][tid] = 0.f;
			s_touched[tid] = 0u;
		}
		// the next iteration's staging barrier orders these LDS writes before the next consume
	}
}

int launch_blend_bwd(const BlendBwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(blend_bwd_kernel, xcd_grid(p.tiles), 256, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
