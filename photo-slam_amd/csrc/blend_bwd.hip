// blend_bwd.hip -- backward of the alpha blend: per-pixel loss gradients -> per-instance
// gradients of colour, 2D mean, conic and opacity.  One wave per 8x8 quad, four quads = one 256-thread workgroup per tile.
//
// Per-pixel semantics are renderCUDA's backward (cuda_rasterizer/backward.cu:399-557): walk
// the tile list back to front starting at each pixel's last contributor, recompute alpha,
// un-blend T, and accumulate the nine partial derivatives.
//
// The reference issues nine global float atomics per contributing (pixel, Gaussian) pair
// (backward.cu:523-554).  Here there are none (see blend.h):
//   * entries behind the deepest last-contributor of the tile are never loaded; a segment of 256 list entries is staged once
//     per tile by all 256 threads, and only the entries the forward blend flagged as blended by some quad of the tile
//     (state.h: contrib); every quad-wave then walks the entries flagged for ITS quad;
//   * the per-pair arithmetic is branch-free; 1/(1-alpha) is one v_rcp_f32 shared by the two
//     divisions of the reference;
//   * "everything behind entry j, dotted with the pixel's loss gradient" is ONE running scalar per pixel (below) instead of
//     the reference's three-channel accum_rec;
//   * the nine terms are summed across the 64 pixels of a quad with a butterfly packed from the
//     top through v_permlane32_swap / v_permlane16_swap (wave64.h, 19 VALU) that leaves eight
//     totals in one register (one per 8-lane group) and the ninth as four row sums; ONE ds_add_f32
//     with twelve active lanes adds them to the tile's LDS accumulators, where the four quads
//     of a tile meet;
//   * the two mean2D terms are reduced as sum(dL_dG*G*dx), sum(dL_dG*G*dy); their conic
//     combination (backward.cu:539-546) is linear in them and applied once per Gaussian in
//     preprocess_bwd (partials.h);
//   * at the end of a segment of 256 list entries the workgroup writes every touched entry's
//     nine sums to that instance's private 48-byte slot with plain stores.
//
// The running scalar.  The reference keeps accum_rec = the colour blended BEHIND entry j, normalised by the transmittance
// there (A_{j+1}), and forms dL/dalpha_j = T_j (c_j - A_{j+1}) . dpix - T_final / (1 - alpha_j) (bg . dpix)
// (backward.cu:505-534).  With the un-normalised suffix S_{j+1} = sum_{k > j} c_k alpha_k T_k = T_{j+1} A_{j+1} and
// T_{j+1} = T_j (1 - alpha_j) this is   dL/dalpha_j = T_j (c_j . dpix) - B_{j+1} / (1 - alpha_j),
//   B_{j+1} = S_{j+1} . dpix + T_final (bg . dpix),     B_j = B_{j+1} + (c_j . dpix) alpha_j T_j,
// one scalar recurrence instead of three (six VALU per visit instead of eleven), no division by a small transmittance -- and a
// state that could be re-created at any list position from T before the position and the colour blended behind it.  (Round 6
// built that: the forward blend left both per pixel at every 256- / 1 024-entry boundary and the backward blend ran one workgroup
// per (tile, group of segments) -- parity green, -7 % of this kernel at 500 k @ 1200 x 680 and 2 M @ 640 x 480, nothing elsewhere,
// +4 us in the forward blend and 16 bytes per instance: removed, commit 2e10ed9, EXPERIMENTS.md R6.)
//
#include "blend.h"
#include "kernels.h"

namespace gsr {

constexpr int BWD_SEG = 256;  // list entries accumulated in LDS per segment (9 x 256 floats = 9 KiB); thread i stages entry i of the segment

__global__ void __launch_bounds__(256)
blend_bwd_kernel(const BlendBwdParams p)
{
	static_assert(BWD_SEG == 256, "one thread per entry of a segment");
	__shared__ float4 s_rec[BWD_SEG][3];   // per entry of the segment: (x, y, A', B') (C', opacity, r, g) (b, -, -, -)
	__shared__ float s_acc[9][BWD_SEG];
	__shared__ uint32_t s_slot[BWD_SEG];
	__shared__ uint8_t s_flag[BWD_SEG];    // bit q: quad q of the tile blends the entry (a byte each: 7 workgroups per CU fit the 160 KiB of LDS)
	__shared__ uint32_t s_wmax[4];

	const int tile = tile_assignment((int)blockIdx.x, p.deal);
	if (tile >= p.tiles) return;
	const int tile_x = tile % p.grid_x, tile_y = tile / p.grid_x;
	const int quad = (int)wave_uniform_u32((uint32_t)wave_id());   // scalar: the LDS record address is SGPR arithmetic
	const int l = lane_id(), tid = (int)threadIdx.x;
	const int qx0 = tile_x * TILE + (quad & 1) * 8, qy0 = tile_y * TILE + (quad >> 1) * 8;
	const int px = qx0 + (l & 7), py = qy0 + (l >> 3);
	const bool inside = px < p.W && py < p.H;
	typedef float v2f __attribute__((vector_size(8)));
	const v2f pxy = {(float)px, (float)py};
	const uint2 range = p.ranges[tile];
	const size_t pix = (size_t)py * p.W + px;
	const size_t plane = (size_t)p.H * p.W;

	const uint32_t last_contributor = inside ? p.n_contrib[pix] : 0u;
	// entries at or behind wmax touch no pixel of the quad; bmax: none of the tile
	const uint32_t wmax = wave_uniform_u32(wave_max_u32(last_contributor));
	if (l == 0) s_wmax[quad] = wmax;
	__syncthreads();
	const uint32_t bmax = wave_uniform_u32(max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));

	const float T_final = inside ? p.final_T[pix] : 0.f;
	float dpr = 0.f, dpg = 0.f, dpb = 0.f;
	if (inside) {
		dpr = p.dL_dpix[pix];
		dpg = p.dL_dpix[plane + pix];
		dpb = p.dL_dpix[2 * plane + pix];
	}
	// the pixel's state behind its last contributor: T = the final transmittance, B = (everything blended behind, i.e. the
	// background) . dpix
	float T = T_final;
	float B = T_final * (p.bg[0] * dpr + p.bg[1] * dpg + p.bg[2] * dpb);
	// lanes 0, 8, .., 56 deliver the eight packed totals, lanes 1, 17, 33, 49 the four row sums of the ninth
	// (wave_reduce9_swap_f32): one ds_add_f32 with twelve active lanes
	const bool red_ninth = (l & 15) == 1;
	const bool red_lane = ((l & 7) == 0) || red_ninth;
	const int red_off = (red_ninth ? 8 : wave_swap9_component(l)) * BWD_SEG;
	const v2f dprg = {dpr, dpg};

	const int seg_first = (int)((bmax + BWD_SEG - 1) / BWD_SEG) - 1, seg_last = 0;
	// The segment's records are staged ONCE per tile, by all 256 threads (thread i: list entry seg_lo + i), and only for the entries
	// some quad of the tile blended (the forward blend's flags: a quarter of the entries of a 1080p view, a tenth at 640 x 480 with
	// 2 M Gaussians).  The list entries and flags of the NEXT segment are asked for while this one is walked.
	auto seg_flags = [&](int seg_) -> uint32_t {
		const uint32_t e = (uint32_t)seg_ * BWD_SEG + (uint32_t)tid;
		uint32_t f = 0u;
		if (e < bmax) {
#pragma unroll
			for (int q = 0; q < QUADS_PER_TILE; q++)
				// (behind a quad's deepest last contributor the forward blend may not have walked: no flags were written there)
				if (e < s_wmax[q] && p.contrib[(size_t)q * p.contrib_stride + range.x + e] != 0) f |= 1u << q;
		}
		return f;
	};
	auto seg_gid = [&](int seg_) -> uint32_t {
		const uint32_t e = (uint32_t)seg_ * BWD_SEG + (uint32_t)tid;
		return e < bmax ? p.point_list[range.x + e] : 0u;
	};
	uint32_t flags_next = seg_first >= 0 ? seg_flags(seg_first) : 0u, gid_next = seg_first >= 0 ? seg_gid(seg_first) : 0u;
	for (int seg = seg_first; seg >= seg_last; seg--) {
		const uint32_t seg_lo = (uint32_t)seg * BWD_SEG;
		const uint32_t seg_hi = min(bmax, seg_lo + BWD_SEG);
		for (int i = tid; i < 9 * BWD_SEG; i += 256) (&s_acc[0][0])[i] = 0.f;
		{
			const uint32_t f = flags_next, gid = gid_next;
			uint32_t slot = 0xFFFFFFFFu;
			float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
			if (f) {
				q0 = p.rec[3 * (size_t)gid + 0];
				q1 = p.rec[3 * (size_t)gid + 1];
				q2 = p.rec[3 * (size_t)gid + 2];
			}
			if (seg > seg_last) {   // (asked for in front of the wait for the records)
				flags_next = seg_flags(seg - 1);
				gid_next = seg_gid(seg - 1);
			}
			if (f) {
				s_rec[tid][0] = prescale_q0(q0);
				s_rec[tid][1] = make_float4(prescale_c(q1.x), q1.y, q1.z, q1.w);
				s_rec[tid][2].x = q2.x;
				const uint32_t rlo = __float_as_uint(q2.y), rhi = __float_as_uint(q2.z);
				const uint32_t minx = rlo & 0xFFFFu, miny = rlo >> 16, maxx = rhi & 0xFFFFu;
				slot = __float_as_uint(q2.w) + ((uint32_t)tile_y - miny) * (maxx - minx) + ((uint32_t)tile_x - minx);
			}
			s_slot[tid] = slot;
			s_flag[tid] = (uint8_t)f;
		}
		__syncthreads();

		for (int b = (int)((seg_hi - seg_lo - 1u) >> 6); b >= 0; b--) {
			unsigned long long m = wave_ballot((((uint32_t)s_flag[b * 64 + l] >> quad) & 1u) != 0u);
			const int base = (int)seg_lo + b * 64;
			const float4(*rec_b)[3] = &s_rec[b * 64];
			while (m) {
				const int bit = 63 - __clzll((long long)m);
#ifdef GSR_EMU
				m &= ~(1ull << bit);
#else
				asm volatile("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));   // (one scalar instruction instead of shift + andn2)
#endif
				const uint32_t pos = (uint32_t)(base + bit);
				const float4 g0 = rec_b[bit][0];
				const float4 g1 = rec_b[bit][1];
				const float gb = rec_b[bit][2].x;
				const v2f dxy = (v2f){g0.x, g0.y} - pxy;
				const float dx = dxy[0], dy = dxy[1];
				const float pw = g0.z * dx * dx + g1.x * dy * dy + g0.w * dx * dy;
				const float G = __builtin_amdgcn_exp2f(pw);
				const float alpha = fminf(0.99f, g1.y * G);
				const bool ok = (pos < last_contributor) && !(pw > 0.0f) && !(alpha < 1.0f / 255.0f);
				if (wave_ballot(ok) == 0ull) continue;  // wave-uniform
				const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
				const float Tn = T * rinv;   // the transmittance in FRONT of this entry
				// dL/dalpha = T_j (c_j . dpix) - B_{j+1} / (1 - alpha_j)   (the running scalar: file header)
				const float cdp = g1.z * dpr + g1.w * dpg + gb * dpb;
				const float dL_dalpha = cdp * Tn - B * rinv;
				// lanes that do not blend this entry contribute exact zeros and keep their state
				const float am = ok ? alpha : 0.f;
				const float dLm = ok ? dL_dalpha : 0.f;
				const float dcol = am * Tn;
				// the per-Gaussian constants (opacity, -1/2, W/2, H/2, the conic in the mean2D terms) are applied
				// after the reduction (preprocess_bwd, partials.h); pairs of products ride in v_pk_mul_f32
				const float wG = dLm * G;
				const v2f c01 = dprg * (v2f){dcol, dcol};
				const v2f t = dxy * (v2f){wG, wG};          // sum w dx, sum w dy
				const v2f m56 = dxy * (v2f){t[0], t[0]};    // sum w dx dx, sum w dx dy
				// order of the nine sums in the LDS accumulators: 0 colour r, 1 w dx, 2 w dx dx, 3 colour b, 4 colour g, 5 w dy,
				// 6 w dx dy, 7 w dy dy, 8 w -- chosen so that the halves of each packed product sit four apart: the
				// butterfly's first level then adds (v0, v4) + (v1, v5) and (v2, v6) + (v3, v7) as register pairs
				// without a move (wave_reduce9_swap_f32); the segment write-out below restores the slot order
				float v[9];
				v[0] = c01[0];
				v[4] = c01[1];
				v[1] = t[0];
				v[5] = t[1];
				v[2] = m56[0];
				v[6] = m56[1];
				v[3] = dcol * dpb;
				v[7] = t[1] * dy;
				v[8] = wG;
				T = ok ? Tn : T;
				B += dcol * cdp;   // (dcol is zero where the entry is not blended)
				float packed, ninth_row;
				wave_reduce9_swap_f32(v, packed, ninth_row);
				GSR_OPAQUE_F32(packed);      // keep the last butterfly adds fused with their DPP moves (the compiler otherwise
				GSR_OPAQUE_F32(ninth_row);   // sinks them into the 12-lane branch as mov_dpp + add)
				if (red_lane) atomicAdd(&(&s_acc[0][0])[red_off + ((int)pos - (int)seg_lo)], red_ninth ? ninth_row : packed);
			}
		}
		__syncthreads();

		// write every touched entry of the segment to its instance slot
		for (int i = tid; i < (int)(seg_hi - seg_lo); i += 256) {
			const uint32_t slot = s_slot[i];
			float any = 0.f;
#pragma unroll
			for (int c = 0; c < 9; c++) any += fabsf(s_acc[c][i]);
			if (slot != 0xFFFFFFFFu && any != 0.f) {   // untouched / all-zero entries stay unflagged: the per-Gaussian sum skips them
				p.touched[slot] = 1;
				float4* dst = reinterpret_cast<float4*>(p.partials + (size_t)slot * (4 * SLOT_F4));
				// slot order (partials.h): colour r g b, w dx, w dy, w dx dx, w dx dy, w dy dy, w
				dst[0] = make_float4(s_acc[0][i], s_acc[4][i], s_acc[3][i], s_acc[1][i]);
				dst[1] = make_float4(s_acc[5][i], s_acc[2][i], s_acc[6][i], s_acc[7][i]);
				reinterpret_cast<float*>(dst + 2)[0] = s_acc[8][i];
			}
		}
		if (seg > seg_last) __syncthreads();
	}
}

int launch_blend_bwd(const BlendBwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(blend_bwd_kernel, tile_grid(p.deal), 256, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
