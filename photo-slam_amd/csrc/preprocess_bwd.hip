// preprocess_bwd.hip -- per-Gaussian backward stage, one fused kernel.
//
// Fuses computeCov2DCUDA (cuda_rasterizer/backward.cu:144-274) and the backward
// preprocessCUDA (backward.cu:346-396, with the SH backward :20-139 and the cov3D backward
// :278-341) into one pass over the Gaussians, so dL_dcov3D and the partial dL_dmean3D never
// round-trip through HBM between two launches, and writes EVERY output element (zeros for
// culled Gaussians) so the caller does not need the reference's torch::zeros pass
// (src/rasterize_points.cu:149-157, 300 B/Gaussian).
//
// HBM per Gaussian: culled: 4 read (radius) + (55+3M)*4 written zeros; visible: reads mean 12,
// cov3D 24, conic grad 16, mean2D grad 12, colour grad 12, SH 12K, scale 12, rot 16, clamp 1;
// writes mean3D 12, cov3D 24, SH 12M, scale 12, rot 16.
#include "state.h"
#include "wave64.h"
#include "kernels.h"
#include "shrows.h"

namespace gsr {


__device__ static const float BSH_C0 = 0.28209479177387814f;
__device__ static const float BSH_C1 = 0.4886025119029199f;
__device__ static const float BSH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float BSH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

#ifdef GSR_EXP_NO_SMALL_STORES   // traffic experiment: drop every output but dL_dsh
#define GSR_SMALL_STORE(x) for (int i_ = 0; i_ < 0; i_++) {}
#else
#define GSR_SMALL_STORE(x) x
#endif
constexpr int PRB_THREADS = 128;   // 2 waves x 13 KiB of row staging per workgroup

__global__ void __launch_bounds__(PRB_THREADS)
preprocess_bwd_kernel(const PreprocessBwdParams p)
{
	__shared__ float4 s_rows[PRB_THREADS / 64][64][ROW_F4_PAD];
	__shared__ uint32_t s_list[PRB_THREADS / 64][64];
#ifdef GSR_EXP_LDS_PAD   // occupancy experiment
	__shared__ uint32_t s_pad[GSR_EXP_LDS_PAD / 4];
	if (p.P < 0) s_pad[threadIdx.x] = 1, p.dL_dopacity[0] = (float)s_pad[threadIdx.x ^ 1];
#endif

	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int w = wave_id(), l = lane_id();
	const size_t wave_first = (size_t)(blockIdx.x * blockDim.x) + (size_t)w * 64;
	const bool in_range = idx < p.P;
	const bool vis = in_range && (p.radii[idx] > 0);
	const int M3 = 3 * p.M;
	const bool rows_ok = p.dL_dsh && (M3 == ROW_F4 * 4) && ((reinterpret_cast<uintptr_t>(p.dL_dsh) & 15) == 0) &&
	                     ((reinterpret_cast<uintptr_t>(p.shs) & 15) == 0);
	float* out_sh = (p.dL_dsh && in_range) ? p.dL_dsh + (size_t)idx * M3 : nullptr;

	if (in_range && !vis) {
		// culled: the reference leaves the torch::zeros content
#pragma unroll
		GSR_SMALL_STORE(for (int i = 0; i < 3; i++) p.dL_dmean2D[3 * (size_t)idx + i] = 0.f);
#pragma unroll
		GSR_SMALL_STORE(for (int i = 0; i < 3; i++) p.dL_dcolor[3 * (size_t)idx + i] = 0.f);
		GSR_SMALL_STORE(p.dL_dopacity[idx] = 0.f);
		GSR_SMALL_STORE(if (p.dL_dconic) reinterpret_cast<float4*>(p.dL_dconic)[idx] = make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
		GSR_SMALL_STORE(for (int i = 0; i < 3; i++) p.dL_dmean3D[3 * (size_t)idx + i] = 0.f);
#pragma unroll
		GSR_SMALL_STORE(for (int i = 0; i < 6; i++) p.dL_dcov3D[6 * (size_t)idx + i] = 0.f);
		if (out_sh && !rows_ok)
			for (int i = 0; i < M3; i++) out_sh[i] = 0.f;
		if (p.dL_dscale) {
#pragma unroll
			GSR_SMALL_STORE(for (int i = 0; i < 3; i++) p.dL_dscale[3 * (size_t)idx + i] = 0.f);
			GSR_SMALL_STORE(reinterpret_cast<float4*>(p.dL_drot)[idx] = make_float4(0.f, 0.f, 0.f, 0.f));
		}
	}

	const float* V = p.view;
	const float* Pm = p.proj;
	float mx = 0.f, my = 0.f, mz = 0.f;
	float gmx = 0.f, gmy = 0.f, gmz = 0.f;
	float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	float4 ga0 = make_float4(0.f, 0.f, 0.f, 0.f);

	if (vis) {
		mx = p.means3D[3 * (size_t)idx];
		my = p.means3D[3 * (size_t)idx + 1];
		mz = p.means3D[3 * (size_t)idx + 2];
		// ------------------------------------------------------------------ computeCov2DCUDA, backward.cu:144-274
		float c3[6];
#pragma unroll
		for (int i = 0; i < 6; i++) c3[i] = p.cov3D[6 * (size_t)idx + i];
		// blend-stage gradients of this Gaussian (colour 0..2, mean2D 3..4, conic 5..7, opacity 8)
		const float4* ga = reinterpret_cast<const float4*>(p.grad_acc) + 3 * (size_t)idx;
		ga0 = ga[0];
		const float4 ga1 = ga[1];
		const float ga2x = ga[2].x;
		const float gcx = ga1.y, gcy = ga1.z, gcz = ga1.w;
		GSR_SMALL_STORE(p.dL_dmean2D[3 * (size_t)idx + 0] = ga0.w);
		GSR_SMALL_STORE(p.dL_dmean2D[3 * (size_t)idx + 1] = ga1.x);
		GSR_SMALL_STORE(p.dL_dmean2D[3 * (size_t)idx + 2] = 0.f);
		GSR_SMALL_STORE(p.dL_dcolor[3 * (size_t)idx + 0] = ga0.x);
		GSR_SMALL_STORE(p.dL_dcolor[3 * (size_t)idx + 1] = ga0.y);
		GSR_SMALL_STORE(p.dL_dcolor[3 * (size_t)idx + 2] = ga0.z);
		// raw logit: d sigmoid = o (1 - o), o = the activated opacity kept in the blend record
		if (p.raw_params & GSR_RAW_OPACITY) {
			const float o = p.rec[3 * (size_t)idx + 1].y;
			GSR_SMALL_STORE(p.dL_dopacity[idx] = ga2x * o * (1.0f - o));
		} else {
			GSR_SMALL_STORE(p.dL_dopacity[idx] = ga2x);
		}
		GSR_SMALL_STORE(if (p.dL_dconic) reinterpret_cast<float4*>(p.dL_dconic)[idx] = make_float4(gcx, gcy, 0.f, gcz));
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
#pragma unroll
		GSR_SMALL_STORE(for (int i = 0; i < 6; i++) p.dL_dcov3D[6 * (size_t)idx + i] = dcov[i]);

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

	// ------------------------------------------------------------------ SH backward, backward.cu:20-139
	if (p.shs) {   // wave-uniform
		const int deg = p.D;
		const int nfl = 3 * (deg + 1) * (deg + 1);
		float sh[48];
		if (rows_ok) {
			wave_load_rows(reinterpret_cast<const float4*>(p.shs), wave_first, (nfl + 3) >> 2, vis, s_rows[w], s_list[w]);
			if (vis) {
#pragma unroll
				for (int i = 0; i < 12; i++)
					if (4 * i < nfl) {
						const float4 v = s_rows[w][l][i];
						sh[4 * i] = v.x; sh[4 * i + 1] = v.y; sh[4 * i + 2] = v.z; sh[4 * i + 3] = v.w;
					}
			}
			wave_fence();  // rows are re-used for the gradient below
		} else if (vis) {
			const float* shrow = p.shs + (size_t)idx * M3;
#pragma unroll
			for (int i = 0; i < 48; i++)
				if (i < nfl) sh[i] = shrow[i];
		}
		float dsh[48];
#pragma unroll
		for (int i = 0; i < 48; i++) dsh[i] = 0.f;
		if (vis) {
			const float ox = mx - p.campos[0], oy = my - p.campos[1], oz = mz - p.campos[2];
			const float len = sqrtf(ox * ox + oy * oy + oz * oz);
			const float x = ox / len, y = oy / len, z = oz / len;
			const uint8_t cl = p.clamped[idx];
			float dRGB[3] = {ga0.x, ga0.y, ga0.z};
			dRGB[0] *= (cl & 1) ? 0.f : 1.f;
			dRGB[1] *= (cl & 2) ? 0.f : 1.f;
			dRGB[2] *= (cl & 4) ? 0.f : 1.f;
			float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};  // dRGBdx/dy/dz
#define SHK(k) sh[3 * (k) + ch]
#define DSH(k, val)                                                 \
	{                                                               \
		const float t_ = (val);                                     \
		dsh[3 * (k)] = t_ * dRGB[0];                                \
		dsh[3 * (k) + 1] = t_ * dRGB[1];                            \
		dsh[3 * (k) + 2] = t_ * dRGB[2];                            \
	}
			DSH(0, BSH_C0);
			if (deg > 0) {
				DSH(1, -BSH_C1 * y);
				DSH(2, BSH_C1 * z);
				DSH(3, -BSH_C1 * x);
#pragma unroll
				for (int ch = 0; ch < 3; ch++) {
					ddx[ch] = -BSH_C1 * SHK(3);
					ddy[ch] = -BSH_C1 * SHK(1);
					ddz[ch] = BSH_C1 * SHK(2);
				}
				if (deg > 1) {
					const float xx = x * x, yy = y * y, zz = z * z;
					const float xy = x * y, yz = y * z, xz = x * z;
					DSH(4, BSH_C2[0] * xy);
					DSH(5, BSH_C2[1] * yz);
					DSH(6, BSH_C2[2] * (2.f * zz - xx - yy));
					DSH(7, BSH_C2[3] * xz);
					DSH(8, BSH_C2[4] * (xx - yy));
#pragma unroll
					for (int ch = 0; ch < 3; ch++) {
						ddx[ch] += BSH_C2[0] * y * SHK(4) + BSH_C2[2] * 2.f * -x * SHK(6) + BSH_C2[3] * z * SHK(7) + BSH_C2[4] * 2.f * x * SHK(8);
						ddy[ch] += BSH_C2[0] * x * SHK(4) + BSH_C2[1] * z * SHK(5) + BSH_C2[2] * 2.f * -y * SHK(6) + BSH_C2[4] * 2.f * -y * SHK(8);
						ddz[ch] += BSH_C2[1] * y * SHK(5) + BSH_C2[2] * 2.f * 2.f * z * SHK(6) + BSH_C2[3] * x * SHK(7);
					}
					if (deg > 2) {
						DSH(9, BSH_C3[0] * y * (3.f * xx - yy));
						DSH(10, BSH_C3[1] * xy * z);
						DSH(11, BSH_C3[2] * y * (4.f * zz - xx - yy));
						DSH(12, BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
						DSH(13, BSH_C3[4] * x * (4.f * zz - xx - yy));
						DSH(14, BSH_C3[5] * z * (xx - yy));
						DSH(15, BSH_C3[6] * x * (xx - 3.f * yy));
#pragma unroll
						for (int ch = 0; ch < 3; ch++) {
							ddx[ch] += (BSH_C3[0] * SHK(9) * 3.f * 2.f * xy + BSH_C3[1] * SHK(10) * yz + BSH_C3[2] * SHK(11) * -2.f * xy +
							            BSH_C3[3] * SHK(12) * -3.f * 2.f * xz + BSH_C3[4] * SHK(13) * (-3.f * xx + 4.f * zz - yy) +
							            BSH_C3[5] * SHK(14) * 2.f * xz + BSH_C3[6] * SHK(15) * 3.f * (xx - yy));
							ddy[ch] += (BSH_C3[0] * SHK(9) * 3.f * (xx - yy) + BSH_C3[1] * SHK(10) * xz +
							            BSH_C3[2] * SHK(11) * (-3.f * yy + 4.f * zz - xx) + BSH_C3[3] * SHK(12) * -3.f * 2.f * yz +
							            BSH_C3[4] * SHK(13) * -2.f * xy + BSH_C3[5] * SHK(14) * -2.f * yz +
							            BSH_C3[6] * SHK(15) * -3.f * 2.f * xy);
							ddz[ch] += (BSH_C3[1] * SHK(10) * xy + BSH_C3[2] * SHK(11) * 4.f * 2.f * yz +
							            BSH_C3[3] * SHK(12) * 3.f * (2.f * zz - xx - yy) + BSH_C3[4] * SHK(13) * 4.f * 2.f * xz +
							            BSH_C3[5] * SHK(14) * (xx - yy));
						}
					}
				}
			}
#undef SHK
#undef DSH
			const float dLx = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
			const float dLy = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
			const float dLz = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
			// dnormvdv, auxiliary.h:107-117
			const float sum2 = ox * ox + oy * oy + oz * oz;
			const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
			gmx += ((+sum2 - ox * ox) * dLx - oy * ox * dLy - oz * ox * dLz) * invsum32;
			gmy += (-ox * oy * dLx + (sum2 - oy * oy) * dLy - oz * oy * dLz) * invsum32;
			gmz += (-ox * oz * dLx - oy * oz * dLy + (sum2 - oz * oz) * dLz) * invsum32;
		}
		// the gradient rows (zeros for culled Gaussians) leave the wave as one contiguous 12 KiB burst
		if (rows_ok) {
#pragma unroll
			for (int i = 0; i < 12; i++) s_rows[w][l][i] = make_float4(dsh[4 * i], dsh[4 * i + 1], dsh[4 * i + 2], dsh[4 * i + 3]);
			const long long left = (long long)p.P - (long long)wave_first;
			wave_store_rows(reinterpret_cast<float4*>(p.dL_dsh), wave_first, (int)(left > 64 ? 64 : (left < 0 ? 0 : left)), s_rows[w]);
		} else if (vis) {
			for (int i = 0; i < (M3 < 48 ? M3 : 48); i++) out_sh[i] = dsh[i];
			for (int i = 48; i < M3; i++) out_sh[i] = 0.f;  // M > 16 is not produced by the reference model; keep the row fully written
		}
	}
	if (!vis) return;
	GSR_SMALL_STORE(p.dL_dmean3D[3 * (size_t)idx + 0] = gmx);
	GSR_SMALL_STORE(p.dL_dmean3D[3 * (size_t)idx + 1] = gmy);
	GSR_SMALL_STORE(p.dL_dmean3D[3 * (size_t)idx + 2] = gmz);

	// ------------------------------------------------------------------ cov3D backward, backward.cu:278-341
	if (p.scales) {
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
		GSR_SMALL_STORE(p.dL_dscale[3 * (size_t)idx + 0] = raw_s ? ds0 * sx : ds0);
		GSR_SMALL_STORE(p.dL_dscale[3 * (size_t)idx + 1] = raw_s ? ds1 * sy : ds1);
		GSR_SMALL_STORE(p.dL_dscale[3 * (size_t)idx + 2] = raw_s ? ds2 * sz : ds2);
		D00 *= s0; D01 *= s0; D02 *= s0;
		D10 *= s1; D11 *= s1; D12 *= s1;
		D20 *= s2; D21 *= s2; D22 *= s2;
		float4 dq;
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
		GSR_SMALL_STORE(reinterpret_cast<float4*>(p.dL_drot)[idx] = dq);
	}
}

int launch_preprocess_bwd(const PreprocessBwdParams& p, hipStream_t stream)
{
	GSR_LAUNCH(preprocess_bwd_kernel, div_up(p.P, PRB_THREADS), PRB_THREADS, stream, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
