// blend_fwd.hip -- forward alpha blending, one wave per 8x8 pixel quad.
//
// Per-pixel semantics are renderCUDA's (cuda_rasterizer/forward.cu:261-374): walk the tile's
// depth-sorted list front to back, skip pairs with power > 0 or alpha < 1/255, stop a pixel
// once T*(1-alpha) < 1e-4 (that entry is not blended), write C + T*bg in CHW, final T and
// the index of the last contributor.
//
// Structure: see blend.h.  The per-pair arithmetic is branch-free (select instead of the
// reference's nested continues), the next batch's list entries are prefetched while the current
// batch is blended, and a wave leaves as soon as all its pixels are saturated.
#include "blend.h"
#include "kernels.h"

namespace gsr {

__global__ void __launch_bounds__(64)
blend_fwd_kernel(const BlendFwdParams p)
{
	__shared__ float4 s_rec[64][3];   // per staged entry: (x, y, A', B') (C', opacity, r, g) (b, -, -, -)

	int tile, quad;
	quad_assignment((int)blockIdx.x, p.tiles, tile, quad);
	if (tile >= p.tiles) return;
	const int tile_x = tile % p.grid_x, tile_y = tile / p.grid_x;
	const int l = lane_id();
	const int qx0 = tile_x * TILE + (quad & 1) * 8, qy0 = tile_y * TILE + (quad >> 1) * 8;
	const int px = qx0 + (l & 7), py = qy0 + (l >> 3);
	const bool inside = px < p.W && py < p.H;
	const float pxf = (float)px, pyf = (float)py;
	const uint2 range = p.ranges[tile];
	const int n = (int)(range.y - range.x);

	float T = 1.0f;
	typedef float v2f __attribute__((vector_size(8)));
	v2f Crg = {0.f, 0.f};   // red and green ride in one v_pk_fma_f32
	float Cb = 0.f;
	uint32_t last_contributor = 0;
	// pixel state predicates live as 64-bit lane masks in SGPR pairs; their logic is scalar
	unsigned long long done_m = wave_ballot(!inside);

	uint32_t gid_next = (l < n) ? p.point_list[range.x + (uint32_t)l] : 0u;
	for (int base = 0; base < n; base += 64) {
		if (~done_m == 0ull) break;
		const bool have = base + l < n;
		const uint32_t gid = gid_next;
		const int e_next = base + 64 + l;
		gid_next = (e_next < n) ? p.point_list[range.x + (uint32_t)e_next] : 0u;
		bool keep = false;
		if (have) {
			const float4 q0 = p.rec[3 * (size_t)gid + 0];
			const float4 q1 = p.rec[3 * (size_t)gid + 1];
			const float cb = p.rec[3 * (size_t)gid + 2].x;
			keep = quad_keep(q0, q1, (float)qx0, (float)qy0);
			s_rec[l][0] = prescale_q0(q0);
			s_rec[l][1] = make_float4(prescale_c(q1.x), q1.y, q1.z, q1.w);
			s_rec[l][2].x = cb;
		}
		unsigned long long m = wave_ballot(keep);
		wave_fence();
		unsigned long long contrib_m = 0ull;   // entries of this batch that some pixel of the quad blends (scalar)
		// The visit loop is bound by VALU issue AND sensitive to scalar issue (three more scalar instructions per visit cost
		// 12 us per launch): its control flow is one scalar mask -- the surviving entries not yet visited, emptied when every
		// pixel is saturated -- cleared bit by bit with s_bitset0_b64 (the compiler's m &= m - 1 is three instructions) and
		// tested once per iteration.
		while (m) {
			const int bit = __ffsll((long long)m) - 1;
#ifdef GSR_EMU
			m &= m - 1ull;
#else
			asm volatile("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));
#endif
			const float4 g0 = s_rec[bit][0];
			const float4 g1 = s_rec[bit][1];
			const float gb = s_rec[bit][2].x;
			const float dx = g0.x - pxf, dy = g0.y - pyf;
			const float pw = g0.z * dx * dx + g1.x * dy * dy + g0.w * dx * dy;   // log2(e) * power
			const float alpha = fminf(0.99f, g1.y * __builtin_amdgcn_exp2f(pw));
			const unsigned long long ok_m = wave_ballot(!(pw > 0.0f)) & wave_ballot(!(alpha < 1.0f / 255.0f)) & ~done_m;
			const float test_T = T * (1.f - alpha);
			const unsigned long long below_m = wave_ballot(test_T < 0.0001f);
			const unsigned long long upd_m = ok_m & ~below_m;
#ifdef GSR_EMU
			if (upd_m) contrib_m |= 1ull << bit;
#else
			// contrib_m |= upd_m ? 1 << bit : 0 in three scalar instructions (the compiler's select form takes five)
			asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cbranch_scc0 1f\n\ts_bitset1_b64 %0, %2\n1:" : "+s"(contrib_m) : "s"(upd_m), "s"(bit) : "scc");
#endif
			done_m |= ok_m & below_m;
			const float wgt = mask_select0_f32(upd_m, alpha * T);
			Crg += (v2f){g1.z, g1.w} * (v2f){wgt, wgt};
			Cb += gb * wgt;
			T = mask_select_f32(upd_m, test_T, T);
			last_contributor = mask_select_u32(upd_m, (uint32_t)(base + bit + 1), last_contributor);
			// every pixel saturated: the rest of the batch is not visited
#ifdef GSR_EMU
			m = (~done_m == 0ull) ? 0ull : m;
#else
			asm volatile("s_cmp_eq_u64 %1, -1\n\ts_cselect_b64 %0, 0, %0" : "+s"(m) : "s"(done_m) : "scc");
#endif
		}
		const bool wave_done = ~done_m == 0ull;
		// the backward pass walks the same batches: it visits only the entries flagged here (15 % of the entries that survive the
		// quad rejection blend into no pixel -- alpha below 1/255 at every pixel centre, or every such pixel saturated)
		if (have) p.contrib[(size_t)quad * p.contrib_stride + range.x + (uint32_t)(base + l)] = (uint8_t)((contrib_m >> l) & 1ull);
		if (wave_done) break;
		wave_fence();  // all lanes have read this batch before the next one overwrites the slice
	}

	if (inside) {
		const size_t pix = (size_t)py * p.W + px;
		const size_t plane = (size_t)p.H * p.W;
		p.final_T[pix] = T;
		p.n_contrib[pix] = last_contributor;
		p.out_color[pix] = Crg[0] + T * p.bg[0];
		p.out_color[plane + pix] = Crg[1] + T * p.bg[1];
		p.out_color[2 * plane + pix] = Cb + T * p.bg[2];
	}
}

int launch_blend_fwd(const BlendFwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(blend_fwd_kernel, quad_grid(p.tiles), 64, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
