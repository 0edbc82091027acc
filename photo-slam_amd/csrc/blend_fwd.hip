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
	quad_assignment((int)blockIdx.x, p.deal, tile, quad);
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
#ifdef GSR_EMU
	v2f Crg = {0.f, 0.f};
#else
	float Cr = 0.f, Cg = 0.f;
#endif
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
		// The visit loop is bound by VALU issue AND by scalar issue (one scalar unit serves the CU's four SIMDs: ~4 SIMD cycles
		// per scalar instruction against ~75 of VALU issue per visit; three more scalar instructions per visit cost 12 us per
		// launch): its control flow is one scalar mask, the surviving entries not yet visited, cleared bit by bit with
		// s_bitset0_b64 (the compiler's m &= m - 1 is three instructions) and tested once per iteration.
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
#ifdef GSR_EMU
			const unsigned long long upd_m = ok_m & ~below_m;
			if (upd_m) contrib_m |= 1ull << bit;
#else
			// upd_m = ok_m & ~below_m, and contrib_m |= upd_m ? 1 << bit : 0 off the SCC that s_andn2_b64 leaves (three scalar
			// instructions for both; the compiler's select form of the second alone takes five)
			unsigned long long upd_m;
			asm volatile("s_andn2_b64 %[upd], %[ok], %[below]\n\ts_cbranch_scc0 1f\n\ts_bitset1_b64 %[c], %[bit]\n1:"
			             : [upd] "=&s"(upd_m), [c] "+s"(contrib_m) : [ok] "s"(ok_m), [below] "s"(below_m), [bit] "s"(bit) : "scc");
#endif
			done_m |= ok_m & below_m;
#ifdef GSR_EMU
			const float wgt = mask_select0_f32(upd_m, alpha * T);
			Crg += (v2f){g1.z, g1.w} * (v2f){wgt, wgt};
			Cb += gb * wgt;
			T = mask_select_f32(upd_m, test_T, T);
			last_contributor = mask_select_u32(upd_m, (uint32_t)(base + bit + 1), last_contributor);
#else
			// The state of the pixels that blend this entry is updated UNDER EXEC = upd_m: three v_fmac and two v_mov (17 issue
			// cycles) instead of a multiply-by-select, a packed fma, and two selects each behind a v_mov (27 of the visit's 85);
			// costs two scalar instructions.
			{
				const float wgt = alpha * T;
				const uint32_t contributor = (uint32_t)(base + bit + 1);
				unsigned long long saved_exec;
				asm volatile("s_and_saveexec_b64 %[save], %[upd]\n\t"
				             "v_fmac_f32 %[cr], %[gr], %[w]\n\t"
				             "v_fmac_f32 %[cg], %[gg], %[w]\n\t"
				             "v_fmac_f32 %[cb], %[gbv], %[w]\n\t"
				             "v_mov_b32 %[t], %[tt]\n\t"
				             "v_mov_b32 %[last], %[c]\n\t"
				             "s_mov_b64 exec, %[save]"
				             : [save] "=&s"(saved_exec), [cr] "+v"(Cr), [cg] "+v"(Cg), [cb] "+v"(Cb), [t] "+v"(T), [last] "+v"(last_contributor)
				             : [upd] "s"(upd_m), [gr] "v"(g1.z), [gg] "v"(g1.w), [gbv] "v"(gb), [w] "v"(wgt), [tt] "v"(test_T), [c] "s"(contributor)
				             : "scc");
			}
#endif
			// (No test for "every pixel saturated" here: the rest of the batch then changes nothing -- ok_m excludes the saturated
			// pixels -- and is at most a few dozen visits once per quad; the test cost every visit two scalar instructions.
			// Tried and rejected: reading the NEXT entry's record from LDS while the current one is blended, unrolled by two so
			// that the register sets alternate -- 22 instead of 15 scalar instructions per visit, 0.212 instead of 0.188 ms.)
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
#ifdef GSR_EMU
		const float Cr = Crg[0], Cg = Crg[1];
#endif
		p.out_color[pix] = Cr + T * p.bg[0];
		p.out_color[plane + pix] = Cg + T * p.bg[1];
		p.out_color[2 * plane + pix] = Cb + T * p.bg[2];
	}
}

int launch_blend_fwd(const BlendFwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(blend_fwd_kernel, quad_grid(p.deal), 64, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
