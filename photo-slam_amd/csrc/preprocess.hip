// preprocess.hip -- per-Gaussian forward stage: cull, project, 3D->2D covariance (EWA),
// conic, screen radius, tile rectangle, SH -> RGB, packed blend record.
//
// Semantics follow preprocessCUDA (cuda_rasterizer/forward.cu:155-256) and its helpers
// (forward.cu:20-152, auxiliary.h:41-164).  This translation unit is compiled with
// -ffp-contract=off and evaluates every expression in the reference's source order, so
// that radii / tile rectangles / tiles_touched are bit-comparable with the CPU oracle
// (nvcc, hipcc and gcc differ in FMA contraction by default).
//
// HBM traffic per Gaussian: reads 12 (mean) [+ 12 scale + 16 rot + 4 opacity + 12*K SH when
// visible]; writes 12 (depth key, tiles, radius) [+ 48 record + 24 cov3D + 8 rect + 1 when
// visible].  One thread per Gaussian; SH rows are read as 16-byte vectors.
#include "state.h"
#include "wave64.h"
#include "kernels.h"
#include "shrows.h"

namespace gsr {


// float -> int with the hardware's semantics (v_cvt_i32_f32: truncate, saturate, NaN -> 0)
__device__ __forceinline__ int f2i(float f)
{
	if (f != f) return 0;
	if (f >= 2147483648.0f) return 2147483647;
	if (f <= -2147483648.0f) return (-2147483647 - 1);
	return (int)f;
}

// auxiliary.h:41-44 (double because of the 1.0 / 0.5 literals)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

#ifndef GSR_FWD_STAGE_ROWS
#define GSR_FWD_STAGE_ROWS STAGE_ROWS
#endif
constexpr int FWD_ROWS = GSR_FWD_STAGE_ROWS;   // SH rows staged per pass and wave
// (PRE_THREADS = 128, state.h: 2 waves per workgroup)

__global__ void __launch_bounds__(PRE_THREADS)
preprocess_fwd_kernel(const PreprocessParams p, GeometryState g)
{
	__shared__ float4 s_rows[PRE_THREADS / 64][FWD_ROWS][ROW_F4_PAD];
	__shared__ uint32_t s_list[PRE_THREADS / 64][64];
	__shared__ uint32_t s_lag[PRE_THREADS / 64][64];   // lazy SH Adam: steps a listed row is behind (shrows.h)

	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	// the per-tile ranges start from zero (rasterizer_impl.cu:310: a memset of its own there; 65 KB at 1080p)
	for (int t = idx; t < p.tiles; t += (int)(gridDim.x * blockDim.x)) p.ranges[t] = make_uint2(0u, 0u);
	// (the offset scan lists the Gaussians with more than LONG_RUN tiles behind this kernel: its counters start from zero)
	for (int t = idx; t < LONG_LISTS * LONG_COUNT_STRIDE; t += (int)(gridDim.x * blockDim.x)) g.long_counts[t] = 0u;
	const int w = wave_id();
	const size_t wave_first = (size_t)(blockIdx.x * blockDim.x) + (size_t)w * 64;
	const bool in_range = idx < p.P;
	// lazy SH Adam: the row's step count is wanted right behind the geometry phase -- asked for now, it is there by then
	const int row_step = (p.lazy.row_step != nullptr && in_range) ? p.lazy.row_step[idx] : 0;
	uint32_t my_tiles = 0;
	int radius_i = 0;
	uint32_t depth_key = DEPTH_KEY_CULLED;
	float px = 0.f, py = 0.f, pz = 0.f;
	float pix = 0.f, piy = 0.f, conx = 0.f, cony = 0.f, conz = 0.f;
	uint32_t rect_lo = 0, rect_hi = 0;

	// ---------------- phase 1: geometry (per lane)
	if (in_range) {
		do {
			px = p.means3D[3 * idx];
			py = p.means3D[3 * idx + 1];
			pz = p.means3D[3 * idx + 2];
			const float* V = p.view;
			const float* Pm = p.proj;
			// in_frustum, auxiliary.h:139-164
			const float vz = V[2] * px + V[6] * py + V[10] * pz + V[14];
			if (vz <= 0.2f) break;
			// transformPoint4x4 + perspective divide, forward.cu:199-201
			const float hx = Pm[0] * px + Pm[4] * py + Pm[8] * pz + Pm[12];
			const float hy = Pm[1] * px + Pm[5] * py + Pm[9] * pz + Pm[13];
			const float hw = Pm[3] * px + Pm[7] * py + Pm[11] * pz + Pm[15];
			const float p_w = 1.0f / (hw + 0.0000001f);
			const float projx = hx * p_w, projy = hy * p_w;

			// computeCov3D, forward.cu:118-152 (M = S*R with S diagonal: M[c][r] = s_r * R[c][r])
			float c3[6];
			if (p.cov3D_precomp != nullptr) {
#pragma unroll
				for (int i = 0; i < 6; i++) c3[i] = p.cov3D_precomp[6 * (size_t)idx + i];
			} else {
				compute_cov3D(p.scales, p.rotations, (size_t)idx, p.scale_modifier, p.raw_params, c3);   // (kernels.h: shared with the backward pass)
			}

			// computeCov2D, forward.cu:74-113
			float tx = V[0] * px + V[4] * py + V[8] * pz + V[12];
			float ty = V[1] * px + V[5] * py + V[9] * pz + V[13];
			const float tz = vz;
			const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
			const float txtz = tx / tz, tytz = ty / tz;
			tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
			ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
			const float J00 = p.focal_x / tz, J02 = -(p.focal_x * tx) / (tz * tz);
			const float J11 = p.focal_y / tz, J12 = -(p.focal_y * ty) / (tz * tz);
			// W[c][r] = view[4r + c] (glm::mat3 built row-wise from the column-major view matrix)
			// T = W*J: T[0][r] = W[0][r]*J00 + W[2][r]*J02 ; T[1][r] = W[1][r]*J11 + W[2][r]*J12 ; T[2][r] = 0
			// (the zero products of the glm expansion add exact zeros and are dropped)
			const float T00 = V[0] * J00 + V[2] * J02, T01 = V[4] * J00 + V[6] * J02, T02 = V[8] * J00 + V[10] * J02;
			const float T10 = V[1] * J11 + V[2] * J12, T11 = V[5] * J11 + V[6] * J12, T12 = V[9] * J11 + V[10] * J12;
			// A = transpose(T) * transpose(Vrk):  A[c][r] = T[r][0]*Vrk[0][c] + T[r][1]*Vrk[1][c] + T[r][2]*Vrk[2][c]
			const float A00 = T00 * c3[0] + T01 * c3[1] + T02 * c3[2];  // c=0,r=0
			const float A10 = T00 * c3[1] + T01 * c3[3] + T02 * c3[4];  // c=1,r=0
			const float A20 = T00 * c3[2] + T01 * c3[4] + T02 * c3[5];  // c=2,r=0
			const float A01 = T10 * c3[0] + T11 * c3[1] + T12 * c3[2];  // c=0,r=1
			const float A11 = T10 * c3[1] + T11 * c3[3] + T12 * c3[4];  // c=1,r=1
			const float A21 = T10 * c3[2] + T11 * c3[4] + T12 * c3[5];  // c=2,r=1
			// cov = A * T:  cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2]
			const float cov00 = (A00 * T00 + A10 * T01 + A20 * T02) + 0.3f;
			const float cov01 = A01 * T00 + A11 * T01 + A21 * T02;
			const float cov11 = (A01 * T10 + A11 * T11 + A21 * T12) + 0.3f;

			// conic + radius, forward.cu:218-232
			const float det = cov00 * cov11 - cov01 * cov01;
			if (det == 0.0f) break;
			const float det_inv = 1.f / det;
			conx = cov11 * det_inv;
			cony = -cov01 * det_inv;
			conz = cov00 * det_inv;
			const float mid = 0.5f * (cov00 + cov11);
			const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
			const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
			const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
			pix = ndc2pix(projx, p.W);
			piy = ndc2pix(projy, p.H);
			// getRect, auxiliary.h:46-56
			const int mr = f2i(my_radius);
			const int rminx = min(p.grid_x, max(0, f2i((pix - mr) / TILE)));
			const int rminy = min(p.grid_y, max(0, f2i((piy - mr) / TILE)));
			const int rmaxx = min(p.grid_x, max(0, f2i((pix + mr + TILE - 1) / TILE)));
			const int rmaxy = min(p.grid_y, max(0, f2i((piy + mr + TILE - 1) / TILE)));
			const uint32_t tiles = (uint32_t)(rmaxy - rminy) * (uint32_t)(rmaxx - rminx);
			if (tiles == 0) break;
			// GSR_STORE_COV3D only (the parity view of the test-suite): the backward pass recomputes the covariance from scale and
			// rotation (kernels.h: compute_cov3D) -- 24 bytes per visible Gaussian neither written here nor gathered there
			if (p.cov3D_precomp == nullptr && (p.raw_params & GSR_STORE_COV3D)) {
#pragma unroll
				for (int i = 0; i < 6; i++) g.cov3D[6 * (size_t)idx + i] = c3[i];
			}
			depth_key = __float_as_uint(vz);
			radius_i = mr;
			my_tiles = tiles;
			rect_lo = (uint32_t)rminx | ((uint32_t)rminy << 16);
			rect_hi = (uint32_t)rmaxx | ((uint32_t)rmaxy << 16);
		} while (0);
	}
	const bool vis = my_tiles != 0;

	// ---------------- phase 2: colour, forward.cu:238-247 (computeColorFromSH :20-71).  The SH rows of the visible
	// lanes are fetched by the whole wave in passes of STAGE_ROWS rows, ranked by visible lane (shrows.h), and each
	// owner streams its row out of LDS; unaligned row pitches fall back to per-lane loads.
	float cr = 0.f, cg = 0.f, cb = 0.f;
	uint8_t clamp_bits = 0;
	if (p.colors_precomp == nullptr) {
		const int ncoef = (p.D + 1) * (p.D + 1);
		const bool rows_ok = (p.M * 3 == ROW_F4 * 4) && ((reinterpret_cast<uintptr_t>(p.shs) & 15) == 0);
		float rgb[3] = {0.f, 0.f, 0.f};
		// the view direction, formed where a lane evaluates its row (nine products that need not live across the staging)
		auto direction = [&]() {
			const float dx = px - p.campos[0], dy = py - p.campos[1], dz = pz - p.campos[2];
			const float len = sqrtf(dx * dx + dy * dy + dz * dz);
			return sh_dir(dx / len, dy / len, dz / len);
		};
		if (rows_ok) {
			const unsigned long long vmask = wave_ballot(vis);
			const int nvis = __popcll(vmask);
			const int rank = __popcll(vmask & lanemask_lt());
			// lazy SH Adam (gsr_sh_adam_lazy): a visible row that is behind (step - 1) takes its missed zero-gradient steps first
			int lag = 0;
			if (p.lazy.row_step != nullptr && vis) {
				lag = p.lazy.step - 1 - row_step;
				lag = lag < 0 ? 0 : (lag >= p.lazy.window ? p.lazy.window - 1 : lag);
			}
			const bool lagging = wave_ballot(lag > 0) != 0;   // wave-uniform, false on nearly every wave of a steady view
			if (vis) {
				s_list[w][rank] = (uint32_t)lane_id();
				s_lag[w][rank] = (uint32_t)lag;
			}
			wave_fence();
			const int nf4 = lagging ? ROW_F4 : (3 * ncoef + 3) >> 2;   // whole rows where some must be updated
			for (int r0 = 0; r0 < nvis; r0 += FWD_ROWS) {
				const int count = (nvis - r0) < FWD_ROWS ? (nvis - r0) : FWD_ROWS;
				wave_load_listed_rows(reinterpret_cast<const float4*>(p.shs), wave_first, nf4, r0, count, s_rows[w], s_list[w]);
				if (lagging) wave_lazy_catch_up_listed(p.lazy, wave_first, r0, count, s_rows[w], s_list[w], s_lag[w]);
				if (vis && rank >= r0 && rank < r0 + count) sh_row_to_rgb(s_rows[w][rank - r0], ncoef, direction(), rgb);
				wave_fence();  // the next pass overwrites the slice
			}
			if (lag > 0) p.lazy.row_step[idx] = p.lazy.step - 1;
		} else if (vis) {
			const float* sh = p.shs + (size_t)idx * p.M * 3;
			const ShDir d = direction();
#pragma unroll
			for (int k = 0; k < 16; k++) {
				if (k < ncoef) {
#pragma unroll
					for (int ch = 0; ch < 3; ch++) {
						const float t = sh_basis(k, d) * sh[3 * k + ch];
						rgb[ch] = (k == 0) ? t : rgb[ch] + t;
					}
				}
			}
		}
		if (vis) {
			cr = rgb[0] + 0.5f;
			cg = rgb[1] + 0.5f;
			cb = rgb[2] + 0.5f;
			clamp_bits = (uint8_t)((cr < 0 ? 1 : 0) | (cg < 0 ? 2 : 0) | (cb < 0 ? 4 : 0));
			cr = fmaxf(cr, 0.0f);
			cg = fmaxf(cg, 0.0f);
			cb = fmaxf(cb, 0.0f);
		}
	} else if (vis) {
		cr = p.colors_precomp[3 * (size_t)idx];
		cg = p.colors_precomp[3 * (size_t)idx + 1];
		cb = p.colors_precomp[3 * (size_t)idx + 2];
	}

	// ---------------- phase 3: outputs
	if (in_range) {
		if (vis) {
			g.rec[3 * (size_t)idx + 0] = make_float4(pix, piy, conx, cony);
			float opac = p.opacities[idx];
			if (p.raw_params & GSR_RAW_OPACITY) opac = 1.0f / (1.0f + expf(-opac));   // getOpacityActivation, :68-71
			g.rec[3 * (size_t)idx + 1] = make_float4(conz, opac, cr, cg);
			g.rec[3 * (size_t)idx + 2] = make_float4(cb, __uint_as_float(rect_lo), __uint_as_float(rect_hi), 0.f);
			g.clamped[idx] = clamp_bits;
			reinterpret_cast<uint2*>(g.rect)[idx] = make_uint2(rect_lo, rect_hi);
		}
		g.depth_key[idx] = depth_key;
		g.tiles_touched[idx] = my_tiles;
		g.radii[idx] = radius_i;
		if (p.radii_out) p.radii_out[idx] = radius_i;
	}
	// num_rendered = sum of tiles_touched, gsr_last_visible_count() = the number of visible Gaussians: every wave leaves its pair
	// with ONE plain store (a wave that sees nothing stores zeros: the array is written in full, nothing is zeroed beforehand and
	// no atomic is issued); the depth sort's first two launches add the pairs up on the side and write the totals into mapped
	// host memory (sort.hip: RadixHostCount).  Replaces reading back the last element of the scan (rasterizer_impl.cu:281),
	// so the host's wait can overlap the depth sort.
	const unsigned long long wave_ballot_of_visible = wave_ballot(my_tiles != 0u);
	const uint32_t wsum = wave_sum_u32(my_tiles);
	// (the largest depth key of a visible Gaussian rides along: the depth sort runs on 27 bits of key - bits(0.2f) and the host
	// checks, when it reads the counts, that no key needed more -- gsr_api.hip)
	const uint32_t wmaxkey = wave_max_u32(my_tiles != 0u ? depth_key : 0u);
	if (lane_id() == 0) g.wave_counts[(size_t)blockIdx.x * (PRE_THREADS / 64) + w] = make_uint4(wsum, (uint32_t)__popcll(wave_ballot_of_visible), wmaxkey, 0u);
}

// checkFrustum, cuda_rasterizer/rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256)
check_frustum_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (idx >= P) return;
	const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
	const float vz = view[2] * px + view[6] * py + view[10] * pz + view[14];
	present[idx] = vz <= 0.2f ? 0 : 1;
}

int launch_preprocess_fwd(const PreprocessParams& p, const GeometryState& g, hipStream_t stream)
{
	GSR_LAUNCH(preprocess_fwd_kernel, div_up(p.P, PRE_THREADS), PRE_THREADS, stream, p, g);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_check_frustum(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t stream)
{
	GSR_LAUNCH(check_frustum_kernel, div_up(P, 256), 256, stream, P, means3D, view, present);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
