// blend_fwd.hip -- forward alpha blending, one workgroup per 16x16 tile.
//
// Per-pixel semantics are renderCUDA's (cuda_rasterizer/forward.cu:261-374): walk the tile's
// depth-sorted list front to back, skip pairs with power > 0 or alpha < 1/255, stop a pixel
// once T*(1-alpha) < 1e-4 (that entry is not blended), write C + T*bg in CHW, final T and
// the index of the last contributor.
//
// Structure (wave64 / LDS):
//   * batches of 256 list entries are staged through LDS: thread t gathers the 48-byte blend
//     record of entry t (three 16-byte loads) -- one gather instead of the reference's three;
//   * while staging, each thread computes the 4 per-quad rejection bits of its entry; one
//     ballot per quad turns them into 64-bit survivor masks kept in LDS;
//   * each wave owns one 8x8 quad and iterates only over the set bits of its masks (scalar
//     s_ff1 loop, LDS broadcast reads), so whole-wave work is spent only on Gaussians that
//     can reach the quad;
//   * a wave leaves as soon as all its pixels are saturated; the workgroup stops staging when
//     all four waves have left.
#include "blend.h"
#include "kernels.h"

namespace gsr {


__global__ void __launch_bounds__(256)
blend_fwd_kernel(const BlendFwdParams p)
{
	__shared__ float4 s_q0[256];
	__shared__ float4 s_q1[256];
	__shared__ float s_b[256];
	__shared__ unsigned long long s_mask[4][4];  // [quad][loader wave]
	__shared__ int s_active[4];

	const int tile = xcd_tile((int)blockIdx.x, p.tiles);
	if (tile >= p.tiles) return;
	const int tile_x = tile % p.grid_x, tile_y = tile / p.grid_x;
	const int w = wave_id(), l = lane_id(), tid = (int)threadIdx.x;
	int px, py;
	quad_pixel(tile_x, tile_y, px, py);
	const bool inside = px < p.W && py < p.H;
	const float pxf = (float)px, pyf = (float)py;
	const uint2 range = p.ranges[tile];
	const int n = (int)(range.y - range.x);
	const int nbatches = (n + 255) >> 8;

	float T = 1.0f;
	float Cr = 0.f, Cg = 0.f, Cb = 0.f;
	uint32_t last_contributor = 0;
	bool done = !inside;

	for (int b = 0; b < nbatches; b++) {
		const bool wave_active = wave_ballot(!done) != 0ull;
		if (l == 0) s_active[w] = wave_active ? 1 : 0;
		__syncthreads();
		if ((s_active[0] | s_active[1] | s_active[2] | s_active[3]) == 0) break;  // workgroup-uniform

		// ---- stage 256 entries
		const int e = (b << 8) + tid;
		uint32_t keep = 0;
		if (e < n) {
			const uint32_t gid = p.point_list[range.x + (uint32_t)e];
			const float4 q0 = p.rec[3 * (size_t)gid + 0];
			const float4 q1 = p.rec[3 * (size_t)gid + 1];
			const float4 q2 = p.rec[3 * (size_t)gid + 2];
			s_q0[tid] = q0;
			s_q1[tid] = q1;
			s_b[tid] = q2.x;
			keep = quad_keep_bits(q0, q1, (float)(tile_x * TILE), (float)(tile_y * TILE));
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const unsigned long long m = wave_ballot((keep >> q) & 1u);
			if (l == 0) s_mask[q][w] = m;
		}
		__syncthreads();

		// ---- consume: wave w blends quad w
		if (wave_active) {
			bool wave_done = false;
			for (int lw = 0; lw < 4 && !wave_done; lw++) {
				unsigned long long m = wave_uniform_u64(s_mask[w][lw]);
				while (m) {
					const int bit = __ffsll((long long)m) - 1;
					m &= m - 1ull;
					const int jj = (lw << 6) + bit;
					const float4 q0 = s_q0[jj];
					const float4 q1 = s_q1[jj];
					const float cb = s_b[jj];
					if (!done) {
						const float dx = q0.x - pxf, dy = q0.y - pyf;
						const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
						if (!(power > 0.0f)) {
							const float alpha = fminf(0.99f, q1.y * __expf(power));
							if (!(alpha < 1.0f / 255.0f)) {
								const float test_T = T * (1.f - alpha);
								if (test_T < 0.0001f) {
									done = true;
								} else {
									const float wgt = alpha * T;
									Cr += q1.z * wgt;
									Cg += q1.w * wgt;
									Cb += cb * wgt;
									T = test_T;
									last_contributor = (uint32_t)((b << 8) + jj + 1);
								}
							}
						}
					}
					if (wave_ballot(!done) == 0ull) {
						wave_done = true;
						break;
					}
				}
			}
		}
	}

	if (inside) {
		const size_t pix = (size_t)py * p.W + px;
		const size_t plane = (size_t)p.H * p.W;
		p.final_T[pix] = T;
		p.n_contrib[pix] = last_contributor;
		p.out_color[pix] = Cr + T * p.bg[0];
		p.out_color[plane + pix] = Cg + T * p.bg[1];
		p.out_color[2 * plane + pix] = Cb + T * p.bg[2];
	}
}

int launch_blend_fwd(const BlendFwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(blend_fwd_kernel, xcd_grid(p.tiles), 256, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
