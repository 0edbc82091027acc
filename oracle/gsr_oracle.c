/*
 * gsr_oracle.c -- CPU restatement of the Photo-SLAM Gaussian rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).  PARITY UNPINNED by the reference's
 * own tests (it has none); every function cites the reference file:line it follows.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (contraction off so that +,-,*,/,sqrt are single IEEE operations evaluated in
 * the source order of the reference expressions; the HIP preprocess kernel is held
 * to the same rule, which is what makes tile rectangles bit-comparable).
 *
 * glm conventions restated (glm is not available here): mat3 m[c][r] is
 * column-major, glm::mat3(a,b,c,d,e,f,g,h,i) has columns (a,b,c),(d,e,f),(g,h,i),
 * and (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] evaluated
 * left to right (glm/detail/type_mat3x3.inl operator*).
 */
#include "gsr_oracle.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* cuda_rasterizer/config.h:16 */
#define BLOCK_Y 16 /* cuda_rasterizer/config.h:17 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                              -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                              0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                              -0.5900435899266435f};

static int g_threads = 0;
void gsro_set_threads(int t) { g_threads = t; }
int gsro_get_threads(void)
{
#ifdef _OPENMP
	return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
	return 1;
#endif
}
#ifdef _OPENMP
#define NT() (g_threads > 0 ? g_threads : omp_get_max_threads())
#else
#define NT() 1
#endif

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { float m[3][3]; } mat3; /* m[c][r] */

static mat3 mat3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
	mat3 r = {{{a, b, c}, {d, e, f}, {g, h, i}}};
	return r;
}
static mat3 mat3_mul(mat3 A, mat3 B)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++)
			R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
	return R;
}
static mat3 mat3_t(mat3 A)
{
	mat3 R;
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++)
			R.m[c][r] = A.m[r][c];
	return R;
}
static float fmin2(float a, float b) { return fminf(a, b); }
static float fmax2(float a, float b) { return fmaxf(a, b); }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
/* float -> int as the device does it (cvt.rzi.s32.f32 / v_cvt_i32_f32): truncate,
 * saturate, NaN -> 0.  Plain C conversion is UB out of range. */
static int f2i_sat(float f)
{
	if (f != f) return 0;
	if (f >= 2147483648.0f) return INT_MAX;
	if (f <= -2147483648.0f) return INT_MIN;
	return (int)f;
}

/* auxiliary.h:41-44 -- evaluated in double because of the 1.0 / 0.5 literals. */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static void getRect(float px, float py, int max_radius, uint32_t* rmin, uint32_t* rmax, int gx, int gy)
{
	rmin[0] = (uint32_t)imin(gx, imax(0, f2i_sat((px - max_radius) / BLOCK_X)));
	rmin[1] = (uint32_t)imin(gy, imax(0, f2i_sat((py - max_radius) / BLOCK_Y)));
	rmax[0] = (uint32_t)imin(gx, imax(0, f2i_sat((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
	rmax[1] = (uint32_t)imin(gy, imax(0, f2i_sat((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:58-77 */
static f3 transformPoint4x3(f3 p, const float* m)
{
	f3 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	        m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
	return t;
}
static f4 transformPoint4x4(f3 p, const float* m)
{
	f4 t = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
	        m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
	return t;
}
/* auxiliary.h:89-97 */
static f3 transformVec4x3Transpose(f3 p, const float* m)
{
	f3 t = {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
	        m[8] * p.x + m[9] * p.y + m[10] * p.z};
	return t;
}
/* auxiliary.h:107-117 */
static f3 dnormvdv3(f3 v, f3 dv)
{
	float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	f3 r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}

/* auxiliary.h:139-164 (the prefiltered printf/__trap branch is a debug guard) */
static int in_frustum(int idx, const float* orig_points, const float* viewmatrix, const float* projmatrix,
                      f3* p_view)
{
	f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
	*p_view = transformPoint4x3(p_orig, viewmatrix);
	(void)projmatrix; /* p_proj is computed and discarded in the reference */
	if (p_view->z <= 0.2f) return 0;
	return 1;
}

/* forward.cu:20-71 */
static f3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                             const float* shs, uint8_t* clamped)
{
	f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	f3 dir = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z); /* glm::length */
	dir.x = dir.x / len;
	dir.y = dir.y / len;
	dir.z = dir.z / len;
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float res[3];
	for (int ch = 0; ch < 3; ch++) {
#define SH(k) sh[3 * (k) + ch]
		float result = SH_C0 * SH(0);
		if (deg > 0) {
			float x = dir.x, y = dir.y, z = dir.z;
			result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
			if (deg > 1) {
				float xx = x * x, yy = y * y, zz = z * z;
				float xy = x * y, yz = y * z, xz = x * z;
				result = result + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) +
				         SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) + SH_C2[3] * xz * SH(7) +
				         SH_C2[4] * (xx - yy) * SH(8);
				if (deg > 2) {
					result = result + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
					         SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
					         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
					         SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
					         SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
				}
			}
		}
#undef SH
		result += 0.5f;
		res[ch] = result;
	}
	clamped[3 * idx + 0] = (res[0] < 0);
	clamped[3 * idx + 1] = (res[1] < 0);
	clamped[3 * idx + 2] = (res[2] < 0);
	f3 out = {fmax2(res[0], 0.0f), fmax2(res[1], 0.0f), fmax2(res[2], 0.0f)};
	return out;
}

/* Shared by forward.cu:74-113 and backward.cu:166-195: builds J, W, T = W*J, Vrk. */
static void cov2d_parts(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                        const float* viewmatrix, f3* t_out, float* txtz_o, float* tytz_o, mat3* Wm, mat3* Tm,
                        mat3* Vrk)
{
	f3 t = transformPoint4x3(mean, viewmatrix);
	const float limx = 1.3f * tan_fovx;
	const float limy = 1.3f * tan_fovy;
	const float txtz = t.x / t.z;
	const float tytz = t.y / t.z;
	t.x = fmin2(limx, fmax2(-limx, txtz)) * t.z;
	t.y = fmin2(limy, fmax2(-limy, tytz)) * t.z;
	mat3 J = mat3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
	                   -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
	*Wm = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8], viewmatrix[1], viewmatrix[5], viewmatrix[9],
	                viewmatrix[2], viewmatrix[6], viewmatrix[10]);
	*Tm = mat3_mul(*Wm, J);
	*Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
	*t_out = t;
	*txtz_o = txtz;
	*tytz_o = tytz;
}

/* forward.cu:74-113 */
static f3 computeCov2D(f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, const float* cov3D,
                       const float* viewmatrix)
{
	f3 t;
	float a, b;
	mat3 W, T, Vrk;
	cov2d_parts(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &a, &b, &W, &T, &Vrk);
	mat3 cov = mat3_mul(mat3_mul(mat3_t(T), mat3_t(Vrk)), T);
	cov.m[0][0] += 0.3f;
	cov.m[1][1] += 0.3f;
	f3 r = {cov.m[0][0], cov.m[0][1], cov.m[1][1]};
	return r;
}

/* Rotation matrix exactly as forward.cu:135-139 / backward.cu:288-292 build it. */
static mat3 quat_R(const float* rot)
{
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3]; /* NOT normalised (forward.cu:127) */
	return mat3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
	                 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
	                 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

/* forward.cu:118-152 */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
	mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
	S.m[0][0] = mod * scale[0];
	S.m[1][1] = mod * scale[1];
	S.m[2][2] = mod * scale[2];
	mat3 R = quat_R(rot);
	mat3 M = mat3_mul(S, R);
	mat3 Sigma = mat3_mul(mat3_t(M), M);
	cov3D[0] = Sigma.m[0][0];
	cov3D[1] = Sigma.m[0][1];
	cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1];
	cov3D[4] = Sigma.m[1][2];
	cov3D[5] = Sigma.m[2][2];
}

/* rasterizer_impl.cu:35-50 */
uint32_t gsro_higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb)
			msb += step;
		else
			msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

/* forward.cu:155-256 preprocessCUDA<3> */
static void preprocess_one(gsro_state* st, int idx, const float* orig_points, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                           float focal_x, float focal_y, int* radii)
{
	const int W = st->W, H = st->H;
	radii[idx] = 0;
	st->tiles_touched[idx] = 0;
	f3 p_view;
	if (!in_frustum(idx, orig_points, viewmatrix, projmatrix, &p_view)) return;

	f3 p_orig = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
	f4 p_hom = transformPoint4x4(p_orig, projmatrix);
	float p_w = 1.0f / (p_hom.w + 0.0000001f);
	f3 p_proj = {p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w};

	const float* cov3D;
	if (cov3D_precomp != NULL) {
		cov3D = cov3D_precomp + (size_t)idx * 6;
	} else {
		computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx,
		             st->cov3D + (size_t)idx * 6);
		cov3D = st->cov3D + (size_t)idx * 6;
	}
	f3 cov = computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix);

	float det = (cov.x * cov.z - cov.y * cov.y);
	if (det == 0.0f) return;
	float det_inv = 1.f / det;
	f3 conic = {cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv};

	float mid = 0.5f * (cov.x + cov.z);
	float lambda1 = mid + sqrtf(fmax2(0.1f, mid * mid - det));
	float lambda2 = mid - sqrtf(fmax2(0.1f, mid * mid - det));
	float my_radius = ceilf(3.f * sqrtf(fmax2(lambda1, lambda2)));
	float pix = ndc2Pix(p_proj.x, W), piy = ndc2Pix(p_proj.y, H);
	uint32_t rmin[2], rmax[2];
	getRect(pix, piy, f2i_sat(my_radius), rmin, rmax, st->grid_x, st->grid_y);
	if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return;

	if (colors_precomp == NULL) {
		f3 c = computeColorFromSH(idx, st->D, st->M, orig_points, cam_pos, shs, st->clamped);
		st->rgb[3 * idx + 0] = c.x;
		st->rgb[3 * idx + 1] = c.y;
		st->rgb[3 * idx + 2] = c.z;
	}
	st->depths[idx] = p_view.z;
	radii[idx] = f2i_sat(my_radius);
	st->means2D[2 * idx] = pix;
	st->means2D[2 * idx + 1] = piy;
	st->conic_opacity[4 * idx + 0] = conic.x;
	st->conic_opacity[4 * idx + 1] = conic.y;
	st->conic_opacity[4 * idx + 2] = conic.z;
	st->conic_opacity[4 * idx + 3] = opacities[idx];
	st->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
}

/* Stable LSD radix sort of (u64 key, u32 value) on bits [0,end_bit): the
 * observable contract of cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:303). */
static void radix_sort_pairs(uint64_t* k_in, uint32_t* v_in, uint64_t* k_out, uint32_t* v_out, size_t n,
                             int end_bit)
{
	uint64_t* ka = k_in;
	uint32_t* va = v_in;
	uint64_t* kb = (uint64_t*)malloc(n * sizeof(uint64_t) + 8);
	uint32_t* vb = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
	uint64_t* ktmp = (uint64_t*)malloc(n * sizeof(uint64_t) + 8);
	uint32_t* vtmp = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
	memcpy(ktmp, ka, n * sizeof(uint64_t));
	memcpy(vtmp, va, n * sizeof(uint32_t));
	uint64_t* src_k = ktmp;
	uint32_t* src_v = vtmp;
	uint64_t* dst_k = kb;
	uint32_t* dst_v = vb;
	for (int shift = 0; shift < end_bit; shift += 8) {
		int bits = end_bit - shift < 8 ? end_bit - shift : 8;
		uint32_t mask = (1u << bits) - 1;
		size_t hist[257] = {0};
		for (size_t i = 0; i < n; i++) hist[((src_k[i] >> shift) & mask) + 1]++;
		for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
		for (size_t i = 0; i < n; i++) {
			size_t p = hist[(src_k[i] >> shift) & mask]++;
			dst_k[p] = src_k[i];
			dst_v[p] = src_v[i];
		}
		uint64_t* tk = src_k; src_k = dst_k; dst_k = tk;
		uint32_t* tv = src_v; src_v = dst_v; dst_v = tv;
	}
	memcpy(k_out, src_k, n * sizeof(uint64_t));
	memcpy(v_out, src_v, n * sizeof(uint32_t));
	free(kb); free(vb); free(ktmp); free(vtmp);
}

/* forward.cu:261-374 renderCUDA<3>, one pixel at a time (the per-pixel arithmetic
 * and its order are exactly the thread's; the cooperative fetch is irrelevant). */
static void render_tile_fwd(const gsro_state* st, int tx, int ty, const float* features, const float* bg,
                            float* out_color)
{
	const int W = st->W, H = st->H;
	const uint32_t rs = st->ranges[2 * (ty * st->grid_x + tx)], re = st->ranges[2 * (ty * st->grid_x + tx) + 1];
	for (int ly = 0; ly < BLOCK_Y; ly++)
		for (int lx = 0; lx < BLOCK_X; lx++) {
			int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
			if (!(px < W && py < H)) continue;
			int pix_id = W * py + px;
			float pixfx = (float)px, pixfy = (float)py;
			float T = 1.0f;
			uint32_t contributor = 0, last_contributor = 0;
			float C[3] = {0, 0, 0};
			uint8_t frag = 0;
			for (uint32_t k = rs; k < re; k++) {
				contributor++;
				uint32_t g = st->point_list[k];
				float dx = st->means2D[2 * g] - pixfx, dy = st->means2D[2 * g + 1] - pixfy;
				const float* co = st->conic_opacity + 4 * (size_t)g;
				float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
				{
					float mag = fabsf(0.5f * co[0] * dx * dx) + fabsf(0.5f * co[2] * dy * dy) + fabsf(co[1] * dx * dy);
					if (fabsf(power) <= 1e-5f * mag + 1e-30f && co[3] >= 1.0f / 255.0f * 0.999f) frag = 1;
				}
				if (power > 0.0f) continue;
				float alpha = fmin2(0.99f, co[3] * expf(power));
				if (fabsf(alpha * 255.0f - 1.0f) <= 1e-4f) frag = 1;
				if (alpha < 1.0f / 255.0f) continue;
				float test_T = T * (1 - alpha);
				if (fabsf(test_T * 10000.0f - 1.0f) <= 2e-4f) frag = 1;
				if (test_T < 0.0001f) break; /* done=true: thread stops rasterizing */
				for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)g + ch] * alpha * T;
				T = test_T;
				last_contributor = contributor;
			}
			st->final_T[pix_id] = T;
			st->n_contrib[pix_id] = last_contributor;
			st->fragile[pix_id] = frag;
			for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
		}
}

static void* xcalloc(size_t n, size_t sz)
{
	void* p = calloc(n ? n : 1, sz);
	if (!p) { fprintf(stderr, "gsr_oracle: out of memory\n"); abort(); }
	return p;
}

void gsro_free(gsro_state* st)
{
	if (!st) return;
	free(st->depths); free(st->clamped); free(st->radii); free(st->means2D); free(st->cov3D);
	free(st->conic_opacity); free(st->rgb); free(st->tiles_touched); free(st->point_offsets);
	free(st->keys_unsorted); free(st->vals_unsorted); free(st->keys_sorted); free(st->point_list);
	free(st->ranges); free(st->final_T); free(st->n_contrib); free(st->fragile);
	free(st);
}

/* rasterizer_impl.cu:198-336 */
gsro_state* gsro_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                         int* radii)
{
	(void)prefiltered;
	gsro_state* st = (gsro_state*)xcalloc(1, sizeof(gsro_state));
	st->P = P; st->D = D; st->M = M; st->W = W; st->H = H;
	st->grid_x = (W + BLOCK_X - 1) / BLOCK_X;
	st->grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
	const int T = st->grid_x * st->grid_y;
	const float focal_y = H / (2.0f * tan_fovy);
	const float focal_x = W / (2.0f * tan_fovx);
	st->depths = xcalloc(P, 4); st->clamped = xcalloc(3 * (size_t)P, 1); st->radii = xcalloc(P, 4);
	st->means2D = xcalloc(2 * (size_t)P, 4); st->cov3D = xcalloc(6 * (size_t)P, 4);
	st->conic_opacity = xcalloc(4 * (size_t)P, 4); st->rgb = xcalloc(3 * (size_t)P, 4);
	st->tiles_touched = xcalloc(P, 4); st->point_offsets = xcalloc(P, 4);
	st->ranges = xcalloc(2 * (size_t)T, 4);
	st->final_T = xcalloc((size_t)W * H, 4); st->n_contrib = xcalloc((size_t)W * H, 4);
	st->fragile = xcalloc((size_t)W * H, 1);
	if (radii == NULL) radii = st->radii;

#pragma omp parallel for schedule(static) num_threads(NT())
	for (int i = 0; i < P; i++)
		preprocess_one(st, i, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
		               colors_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, focal_x, focal_y, radii);
	if (radii != st->radii) memcpy(st->radii, radii, (size_t)P * 4);

	/* InclusiveSum, rasterizer_impl.cu:276 */
	uint32_t acc = 0;
	for (int i = 0; i < P; i++) { acc += st->tiles_touched[i]; st->point_offsets[i] = acc; }
	const uint32_t R = P > 0 ? st->point_offsets[P - 1] : 0;
	st->R = (int)R;
	st->keys_unsorted = xcalloc(R, 8); st->vals_unsorted = xcalloc(R, 4);
	st->keys_sorted = xcalloc(R, 8); st->point_list = xcalloc(R, 4);

	/* duplicateWithKeys, rasterizer_impl.cu:70-111 */
#pragma omp parallel for schedule(static) num_threads(NT())
	for (int idx = 0; idx < P; idx++) {
		if (radii[idx] > 0) {
			uint32_t off = (idx == 0) ? 0 : st->point_offsets[idx - 1];
			uint32_t rmin[2], rmax[2];
			getRect(st->means2D[2 * idx], st->means2D[2 * idx + 1], radii[idx], rmin, rmax, st->grid_x, st->grid_y);
			for (int y = (int)rmin[1]; y < (int)rmax[1]; y++)
				for (int x = (int)rmin[0]; x < (int)rmax[0]; x++) {
					uint64_t key = (uint64_t)(y * st->grid_x + x);
					key <<= 32;
					uint32_t dbits;
					memcpy(&dbits, &st->depths[idx], 4);
					key |= dbits;
					st->keys_unsorted[off] = key;
					st->vals_unsorted[off] = (uint32_t)idx;
					off++;
				}
		}
	}
	int bit = (int)gsro_higher_msb((uint32_t)T);
	st->sort_bits = 32 + bit;
	radix_sort_pairs(st->keys_unsorted, st->vals_unsorted, st->keys_sorted, st->point_list, R, 32 + bit);

	/* identifyTileRanges, rasterizer_impl.cu:116-138 (ranges memset at :310) */
	for (uint32_t i = 0; i < R; i++) {
		uint32_t currtile = (uint32_t)(st->keys_sorted[i] >> 32);
		if (i == 0)
			st->ranges[2 * currtile] = 0;
		else {
			uint32_t prevtile = (uint32_t)(st->keys_sorted[i - 1] >> 32);
			if (currtile != prevtile) {
				st->ranges[2 * prevtile + 1] = i;
				st->ranges[2 * currtile] = i;
			}
		}
		if (i == R - 1) st->ranges[2 * currtile + 1] = R;
	}

	const float* feature_ptr = colors_precomp != NULL ? colors_precomp : st->rgb;
#pragma omp parallel for schedule(dynamic, 1) num_threads(NT())
	for (int t = 0; t < T; t++) render_tile_fwd(st, t % st->grid_x, t / st->grid_x, feature_ptr, background, out_color);
	return st;
}

static inline void acc_add(double* p, double v)
{
#pragma omp atomic
	*p += v;
}

/* backward.cu:399-557 renderCUDA<3> (backward), one pixel at a time.  acc is
 * [P][9] doubles: color rgb, mean2D xy, conic x,y,w, opacity. */
static void render_tile_bwd(const gsro_state* st, int tx, int ty, const float* bg_color, const float* colors,
                            const float* dL_dpixels, double* acc)
{
	const int W = st->W, H = st->H;
	const uint32_t rs = st->ranges[2 * (ty * st->grid_x + tx)], re = st->ranges[2 * (ty * st->grid_x + tx) + 1];
	const float ddelx_dx = (float)(0.5 * W);
	const float ddely_dy = (float)(0.5 * H);
	for (int ly = 0; ly < BLOCK_Y; ly++)
		for (int lx = 0; lx < BLOCK_X; lx++) {
			int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
			if (!(px < W && py < H)) continue;
			int pix_id = W * py + px;
			const float pixfx = (float)px, pixfy = (float)py;
			const float T_final = st->final_T[pix_id];
			float T = T_final;
			uint32_t contributor = re - rs;
			const int last_contributor = (int)st->n_contrib[pix_id];
			float accum_rec[3] = {0, 0, 0};
			float dL_dpixel[3];
			for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
			float last_alpha = 0;
			float last_color[3] = {0, 0, 0};
			for (uint32_t k = re; k-- > rs;) {
				contributor--;
				if (contributor >= (uint32_t)last_contributor) continue;
				const uint32_t g = st->point_list[k];
				const float dx = st->means2D[2 * g] - pixfx, dy = st->means2D[2 * g + 1] - pixfy;
				const float* co = st->conic_opacity + 4 * (size_t)g;
				const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
				if (power > 0.0f) continue;
				const float G = expf(power);
				const float alpha = fmin2(0.99f, co[3] * G);
				if (alpha < 1.0f / 255.0f) continue;
				T = T / (1.f - alpha);
				const float dchannel_dcolor = alpha * T;
				float dL_dalpha = 0.0f;
				double* a = acc + 9 * (size_t)g;
				for (int ch = 0; ch < 3; ch++) {
					const float c = colors[3 * (size_t)g + ch];
					accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
					last_color[ch] = c;
					const float dL_dchannel = dL_dpixel[ch];
					dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
					acc_add(&a[ch], (double)(dchannel_dcolor * dL_dchannel));
				}
				dL_dalpha *= T;
				last_alpha = alpha;
				float bg_dot_dpixel = 0;
				for (int i = 0; i < 3; i++) bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
				dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
				const float dL_dG = co[3] * dL_dalpha;
				const float gdx = G * dx;
				const float gdy = G * dy;
				const float dG_ddelx = -gdx * co[0] - gdy * co[1];
				const float dG_ddely = -gdy * co[2] - gdx * co[1];
				acc_add(&a[3], (double)(dL_dG * dG_ddelx * ddelx_dx));
				acc_add(&a[4], (double)(dL_dG * dG_ddely * ddely_dy));
				acc_add(&a[5], (double)(-0.5f * gdx * dx * dL_dG));
				acc_add(&a[6], (double)(-0.5f * gdx * dy * dL_dG));
				acc_add(&a[7], (double)(-0.5f * gdy * dy * dL_dG));
				acc_add(&a[8], (double)(G * dL_dalpha));
			}
		}
}

/* backward.cu:144-274 computeCov2DCUDA */
static void computeCov2D_bwd(int idx, const float* means, const int* radii, const float* cov3Ds, float h_x,
                             float h_y, float tan_fovx, float tan_fovy, const float* view_matrix,
                             const float* dL_dconics, float* dL_dmeans, float* dL_dcov)
{
	if (!(radii[idx] > 0)) return;
	const float* cov3D = cov3Ds + 6 * (size_t)idx;
	f3 mean = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	f3 dL_dconic = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
	f3 t;
	float txtz, tytz;
	mat3 W, T, Vrk;
	cov2d_parts(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &t, &txtz, &tytz, &W, &T, &Vrk);
	const float limx = 1.3f * tan_fovx;
	const float limy = 1.3f * tan_fovy;
	const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
	const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
	mat3 cov2D = mat3_mul(mat3_mul(mat3_t(T), mat3_t(Vrk)), T);
	float a = cov2D.m[0][0] += 0.3f;
	float b = cov2D.m[0][1];
	float c = cov2D.m[1][1] += 0.3f;
	float denom = a * c - b * b;
	float dL_da = 0, dL_db = 0, dL_dc = 0;
	float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(c_, r_) T.m[c_][r_]
#define VV(c_, r_) Vrk.m[c_][r_]
#define WW(c_, r_) W.m[c_][r_]
	if (denom2inv != 0) {
		dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
		dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
		dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);
		dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
		dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
		dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
		dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db +
		                       2 * TT(1, 0) * TT(1, 1) * dL_dc;
		dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db +
		                       2 * TT(1, 0) * TT(1, 2) * dL_dc;
		dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db +
		                       2 * TT(1, 1) * TT(1, 2) * dL_dc;
	} else {
		for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
	}
	float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
	                (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
	float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
	                (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
	float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
	                (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
	float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
	                (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
	float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
	                (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
	float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
	                (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
	float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
	float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
	float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
	float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
	float tz = 1.f / t.z;
	float tz2 = tz * tz;
	float tz3 = tz2 * tz;
	float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
	float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
	float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 +
	               (2 * h_y * t.y) * tz3 * dL_dJ12;
	f3 dt = {dL_dtx, dL_dty, dL_dtz};
	f3 dL_dmean = transformVec4x3Transpose(dt, view_matrix);
	dL_dmeans[3 * idx + 0] = dL_dmean.x; /* overwrite, backward.cu:273 */
	dL_dmeans[3 * idx + 1] = dL_dmean.y;
	dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* backward.cu:20-139 computeColorFromSH (backward) */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                   const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                   float* dL_dmeans, float* dL_dshs)
{
	f3 pos = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	f3 dir_orig = {pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]};
	float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
	f3 dir = {dir_orig.x / len, dir_orig.y / len, dir_orig.z / len};
	const float* sh = shs + (size_t)idx * max_coeffs * 3;
	float dL_dRGB[3] = {dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
	dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
	dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
	dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
	float x = dir.x, y = dir.y, z = dir.z;
	float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
	float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(k) sh[3 * (k) + ch]
#define DSH(k, v) for (int ch = 0; ch < 3; ch++) dL_dsh[3 * (k) + ch] = (v) * dL_dRGB[ch]
	float dRGBdsh0 = SH_C0;
	DSH(0, dRGBdsh0);
	if (deg > 0) {
		float dRGBdsh1 = -SH_C1 * y;
		float dRGBdsh2 = SH_C1 * z;
		float dRGBdsh3 = -SH_C1 * x;
		DSH(1, dRGBdsh1);
		DSH(2, dRGBdsh2);
		DSH(3, dRGBdsh3);
		for (int ch = 0; ch < 3; ch++) {
			dRGBdx[ch] = -SH_C1 * SH(3);
			dRGBdy[ch] = -SH_C1 * SH(1);
			dRGBdz[ch] = SH_C1 * SH(2);
		}
		if (deg > 1) {
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			float dRGBdsh4 = SH_C2[0] * xy;
			float dRGBdsh5 = SH_C2[1] * yz;
			float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
			float dRGBdsh7 = SH_C2[3] * xz;
			float dRGBdsh8 = SH_C2[4] * (xx - yy);
			DSH(4, dRGBdsh4);
			DSH(5, dRGBdsh5);
			DSH(6, dRGBdsh6);
			DSH(7, dRGBdsh7);
			DSH(8, dRGBdsh8);
			for (int ch = 0; ch < 3; ch++) {
				dRGBdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) +
				              SH_C2[4] * 2.f * x * SH(8);
				dRGBdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) +
				              SH_C2[4] * 2.f * -y * SH(8);
				dRGBdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
			}
			if (deg > 2) {
				float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
				float dRGBdsh10 = SH_C3[1] * xy * z;
				float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
				float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
				float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
				float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
				DSH(9, dRGBdsh9);
				DSH(10, dRGBdsh10);
				DSH(11, dRGBdsh11);
				DSH(12, dRGBdsh12);
				DSH(13, dRGBdsh13);
				DSH(14, dRGBdsh14);
				DSH(15, dRGBdsh15);
				for (int ch = 0; ch < 3; ch++) {
					dRGBdx[ch] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz +
					               SH_C3[2] * SH(11) * -2.f * xy + SH_C3[3] * SH(12) * -3.f * 2.f * xz +
					               SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14) * 2.f * xz +
					               SH_C3[6] * SH(15) * 3.f * (xx - yy));
					dRGBdy[ch] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
					               SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
					               SH_C3[3] * SH(12) * -3.f * 2.f * yz + SH_C3[4] * SH(13) * -2.f * xy +
					               SH_C3[5] * SH(14) * -2.f * yz + SH_C3[6] * SH(15) * -3.f * 2.f * xy);
					dRGBdz[ch] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
					               SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
					               SH_C3[4] * SH(13) * 4.f * 2.f * xz + SH_C3[5] * SH(14) * (xx - yy));
				}
			}
		}
	}
#undef SH
#undef DSH
	/* glm::dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z */
	f3 dL_ddir = {dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
	              dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
	              dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
	f3 dL_dmean = dnormvdv3(dir_orig, dL_ddir);
	dL_dmeans[3 * idx + 0] += dL_dmean.x;
	dL_dmeans[3 * idx + 1] += dL_dmean.y;
	dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* backward.cu:278-341 computeCov3D (backward) */
static void computeCov3D_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                             float* dL_dscales, float* dL_drots)
{
	float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
	mat3 R = quat_R(rot);
	mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
	float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
	S.m[0][0] = s[0];
	S.m[1][1] = s[1];
	S.m[2][2] = s[2];
	mat3 M = mat3_mul(S, R);
	const float* d = dL_dcov3Ds + 6 * (size_t)idx;
	mat3 dL_dSigma = mat3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2],
	                           0.5f * d[4], d[5]);
	/* 2.0f * M * dL_dSigma: scalar*mat first, then mat*mat */
	mat3 M2 = M;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = 2.0f * M.m[c][rr];
	mat3 dL_dM = mat3_mul(M2, dL_dSigma);
	mat3 Rt = mat3_t(R);
	mat3 dL_dMt = mat3_t(dL_dM);
	float* ds = dL_dscales + 3 * (size_t)idx;
	for (int k = 0; k < 3; k++)
		ds[k] = Rt.m[k][0] * dL_dMt.m[k][0] + Rt.m[k][1] * dL_dMt.m[k][1] + Rt.m[k][2] * dL_dMt.m[k][2];
	for (int k = 0; k < 3; k++)
		for (int rr = 0; rr < 3; rr++) dL_dMt.m[k][rr] *= s[k];
#define D(c_, r_) dL_dMt.m[c_][r_]
	f4 q;
	q.x = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
	q.y = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) -
	      4 * x * (D(2, 2) + D(1, 1));
	q.z = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) -
	      4 * y * (D(2, 2) + D(0, 0));
	q.w = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) -
	      4 * z * (D(1, 1) + D(0, 0));
#undef D
	float* dr = dL_drots + 4 * (size_t)idx;
	dr[0] = q.x; dr[1] = q.y; dr[2] = q.z; dr[3] = q.w; /* no normalisation Jacobian, backward.cu:340 */
}

/* backward.cu:346-396 preprocessCUDA<3> (backward) */
static void preprocess_bwd_one(int idx, int D, int M, const float* means, const int* radii, const float* shs,
                               const uint8_t* clamped, const float* scales, const float* rotations,
                               float scale_modifier, const float* proj, const float* campos,
                               const float* dL_dmean2D, float* dL_dmeans, float* dL_dcolor, float* dL_dcov3D,
                               float* dL_dsh, float* dL_dscale, float* dL_drot)
{
	if (!(radii[idx] > 0)) return;
	f3 m = {means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]};
	f4 m_hom = transformPoint4x4(m, proj);
	float m_w = 1.0f / (m_hom.w + 0.0000001f);
	float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
	float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
	float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
	f3 dL_dmean;
	dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
	dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
	dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
	dL_dmeans[3 * idx + 0] += dL_dmean.x;
	dL_dmeans[3 * idx + 1] += dL_dmean.y;
	dL_dmeans[3 * idx + 2] += dL_dmean.z;
	if (shs) computeColorFromSH_bwd(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
	if (scales)
		computeCov3D_bwd(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D,
		                 dL_dscale, dL_drot);
}

/* rasterizer_impl.cu:340-433 */
void gsro_backward(const gsro_state* st, const float* background, const float* means3D, const float* shs,
                   const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                   const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                   const float* cam_pos, float tan_fovx, float tan_fovy, const float* dL_dpix, float* dL_dmean2D,
                   float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                   float* dL_dsh, float* dL_dscale, float* dL_drot)
{
	const int P = st->P, W = st->W, H = st->H;
	const int T = st->grid_x * st->grid_y;
	const float focal_y = H / (2.0f * tan_fovy);
	const float focal_x = W / (2.0f * tan_fovx);
	const int* radii = st->radii;
	const float* color_ptr = (colors_precomp != NULL) ? colors_precomp : st->rgb;
	double* acc = (double*)xcalloc(9 * (size_t)P, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1) num_threads(NT())
	for (int t = 0; t < T; t++) render_tile_bwd(st, t % st->grid_x, t / st->grid_x, background, color_ptr, dL_dpix, acc);
	for (int i = 0; i < P; i++) {
		const double* a = acc + 9 * (size_t)i;
		dL_dcolor[3 * i + 0] += (float)a[0];
		dL_dcolor[3 * i + 1] += (float)a[1];
		dL_dcolor[3 * i + 2] += (float)a[2];
		dL_dmean2D[3 * i + 0] += (float)a[3];
		dL_dmean2D[3 * i + 1] += (float)a[4];
		dL_dconic[4 * i + 0] += (float)a[5];
		dL_dconic[4 * i + 1] += (float)a[6];
		dL_dconic[4 * i + 3] += (float)a[7];
		dL_dopacity[i] += (float)a[8];
	}
	free(acc);
	const float* cov3D_ptr = (cov3D_precomp != NULL) ? cov3D_precomp : st->cov3D;
#pragma omp parallel for schedule(static) num_threads(NT())
	for (int i = 0; i < P; i++)
		computeCov2D_bwd(i, means3D, radii, cov3D_ptr, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, dL_dconic,
		                 dL_dmean3D, dL_dcov3D);
#pragma omp parallel for schedule(static) num_threads(NT())
	for (int i = 0; i < P; i++)
		preprocess_bwd_one(i, st->D, st->M, means3D, radii, shs, st->clamped, scales, rotations, scale_modifier,
		                   projmatrix, cam_pos, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale,
		                   dL_drot);
}

/* rasterizer_impl.cu:54-66,141-153 */
void gsro_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       uint8_t* present)
{
	for (int i = 0; i < P; i++) {
		f3 pv;
		present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, projmatrix, &pv);
	}
}

/* ---------------- simple-knn (third_party/simple-knn/simple_knn.cu) ---------------- */
#define BOX_SIZE 1024
/* simple_knn.cu:45-52 */
static uint32_t prepMorton(uint32_t x)
{
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}
/* float -> uint32 as cvt.rzi.u32.f32 does (NaN -> 0, saturating) */
static uint32_t f2u_sat(float f)
{
	if (!(f > 0.0f)) return 0;
	if (f >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)f;
}
/* simple_knn.cu:54-61 */
static uint32_t coord2Morton(f3 c, f3 mn, f3 mx)
{
	uint32_t x = prepMorton(f2u_sat(((c.x - mn.x) / (mx.x - mn.x)) * ((1 << 10) - 1)));
	uint32_t y = prepMorton(f2u_sat(((c.y - mn.y) / (mx.y - mn.y)) * ((1 << 10) - 1)));
	uint32_t z = prepMorton(f2u_sat(((c.z - mn.z) / (mx.z - mn.z)) * ((1 << 10) - 1)));
	return x | (y << 1) | (z << 2);
}
typedef struct { f3 minn, maxx; } MinMax;
/* simple_knn.cu:119-129 */
static float distBoxPoint(const MinMax* box, f3 p)
{
	f3 diff = {0, 0, 0};
	if (p.x < box->minn.x || p.x > box->maxx.x) diff.x = fmin2(fabsf(p.x - box->minn.x), fabsf(p.x - box->maxx.x));
	if (p.y < box->minn.y || p.y > box->maxx.y) diff.y = fmin2(fabsf(p.y - box->minn.y), fabsf(p.y - box->maxx.y));
	if (p.z < box->minn.z || p.z > box->maxx.z) diff.z = fmin2(fabsf(p.z - box->minn.z), fabsf(p.z - box->maxx.z));
	return diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
}
/* simple_knn.cu:131-145 */
static void updateKBest3(f3 ref, f3 point, float* knn)
{
	f3 d = {point.x - ref.x, point.y - ref.y, point.z - ref.z};
	float dist = d.x * d.x + d.y * d.y + d.z * d.z;
	for (int j = 0; j < 3; j++) {
		if (knn[j] > dist) {
			float t = knn[j];
			knn[j] = dist;
			dist = t;
		}
	}
}
static f3 pt(const float* p, uint32_t i)
{
	f3 r = {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]};
	return r;
}
/* simple_knn.cu:185-221 */
void gsro_knn(int P, const float* points, float* meanDists)
{
	if (P <= 0) return;
	/* cub::DeviceReduce with init {0,0,0} for BOTH min and max (:191-200) */
	f3 mn = {0, 0, 0}, mx = {0, 0, 0};
	for (int i = 0; i < P; i++) {
		f3 p = pt(points, i);
		mn.x = fmin2(mn.x, p.x); mn.y = fmin2(mn.y, p.y); mn.z = fmin2(mn.z, p.z);
		mx.x = fmax2(mx.x, p.x); mx.y = fmax2(mx.y, p.y); mx.z = fmax2(mx.z, p.z);
	}
	uint32_t* morton = (uint32_t*)xcalloc(P, 4);
	uint32_t* idx_a = (uint32_t*)xcalloc(P, 4);
	uint32_t* idx_b = (uint32_t*)xcalloc(P, 4);
	uint32_t* key_a = (uint32_t*)xcalloc(P, 4);
	uint32_t* key_b = (uint32_t*)xcalloc(P, 4);
	for (int i = 0; i < P; i++) {
		morton[i] = coord2Morton(pt(points, i), mn, mx);
		key_a[i] = morton[i];
		idx_a[i] = (uint32_t)i; /* thrust::sequence :207 */
	}
	/* stable u32 radix sort, 32 bits (:210-213) */
	for (int shift = 0; shift < 32; shift += 8) {
		size_t hist[257] = {0};
		for (int i = 0; i < P; i++) hist[((key_a[i] >> shift) & 255) + 1]++;
		for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
		for (int i = 0; i < P; i++) {
			size_t p = hist[(key_a[i] >> shift) & 255]++;
			key_b[p] = key_a[i];
			idx_b[p] = idx_a[i];
		}
		uint32_t* t = key_a; key_a = key_b; key_b = t;
		t = idx_a; idx_a = idx_b; idx_b = t;
	}
	const uint32_t* indices = idx_a;
	const int num_boxes = (P + BOX_SIZE - 1) / BOX_SIZE;
	MinMax* boxes = (MinMax*)xcalloc(num_boxes, sizeof(MinMax));
	/* boxMinMax :78-117 */
	for (int b = 0; b < num_boxes; b++) {
		MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
		for (int i = b * BOX_SIZE; i < imin(P, (b + 1) * BOX_SIZE); i++) {
			f3 p = pt(points, indices[i]);
			me.minn.x = fmin2(me.minn.x, p.x); me.minn.y = fmin2(me.minn.y, p.y); me.minn.z = fmin2(me.minn.z, p.z);
			me.maxx.x = fmax2(me.maxx.x, p.x); me.maxx.y = fmax2(me.maxx.y, p.y); me.maxx.z = fmax2(me.maxx.z, p.z);
		}
		boxes[b] = me;
	}
	/* boxMeanDist :147-183 */
#pragma omp parallel for schedule(dynamic, 256) num_threads(NT())
	for (int idx = 0; idx < P; idx++) {
		f3 point = pt(points, indices[idx]);
		float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
		for (int i = imax(0, idx - 3); i <= imin(P - 1, idx + 3); i++) {
			if (i == idx) continue;
			updateKBest3(point, pt(points, indices[i]), best);
		}
		float reject = best[2];
		best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
		for (int b = 0; b < num_boxes; b++) {
			float dist = distBoxPoint(&boxes[b], point);
			if (dist > reject || dist > best[2]) continue;
			for (int i = b * BOX_SIZE; i < imin(P, (b + 1) * BOX_SIZE); i++) {
				if (i == idx) continue;
				updateKBest3(point, pt(points, indices[i]), best);
			}
		}
		meanDists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
	}
	free(morton); free(idx_a); free(idx_b); free(key_a); free(key_b); free(boxes);
}

void gsro_knn_bruteforce(int P, const float* points, float* meanDists)
{
#pragma omp parallel for schedule(static) num_threads(NT())
	for (int i = 0; i < P; i++) {
		f3 p = pt(points, i);
		float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
		for (int j = 0; j < P; j++) {
			if (j == i) continue;
			updateKBest3(p, pt(points, j), best);
		}
		meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
	}
}


/* ---------------- analysis helper (not part of the reference): work statistics of the
 * per-quad rejection used by the HIP blend kernels (csrc/blend.h quad_keep_bits restated),
 * and a check that it never rejects a pair the reference would blend.
 * out[0] = sum over tiles of list length            out[1] = sum over tiles of max n_contrib (entries staged by bwd)
 * out[2] = (quad, entry) pairs visited by fwd        out[3] = (quad, entry) pairs visited by bwd
 * out[4] = (pixel, entry) pairs blended (alpha>=1/255, before termination)
 * out[5] = (pixel, entry) pairs evaluated by the reference fwd (until done)
 * out[6] = number of blended pairs that the quad test would have rejected (must be 0)
 * out[7] = (quad, entry) pairs without rejection in fwd (4 * entries until the quad is done) */
static uint32_t quad_keep_bits_ref(const float* m2, const float* co, float tile_px0, float tile_py0)
{
	const float mx = m2[0], my = m2[1], A = co[0], B = co[1], C = co[2], o = co[3];
	if (o < 1.0f / 255.0f) return 0u;
	const float det = A * C - B * B;
	if (!(A > 0.f && C > 0.f && det > 0.f)) return 0xFu;
	const float thr = logf(255.0f * o);
	uint32_t bits = 0;
	const float invA = 1.0f / A, invC = 1.0f / C;
	for (int q = 0; q < 4; q++) {
		const float u0 = tile_px0 + (float)((q & 1) * 8) - mx, u1 = u0 + 7.0f;
		const float v0 = tile_py0 + (float)((q >> 1) * 8) - my, v1 = v0 + 7.0f;
		float qmin;
		if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) qmin = 0.f;
		else {
			float e, vs = fminf(v1, fmaxf(v0, -B * u0 * invC));
			qmin = 0.5f * (A * u0 * u0 + C * vs * vs) + B * u0 * vs;
			vs = fminf(v1, fmaxf(v0, -B * u1 * invC));
			e = 0.5f * (A * u1 * u1 + C * vs * vs) + B * u1 * vs; qmin = fminf(qmin, e);
			float us = fminf(u1, fmaxf(u0, -B * v0 * invA));
			e = 0.5f * (A * us * us + C * v0 * v0) + B * us * v0; qmin = fminf(qmin, e);
			us = fminf(u1, fmaxf(u0, -B * v1 * invA));
			e = 0.5f * (A * us * us + C * v1 * v1) + B * us * v1; qmin = fminf(qmin, e);
		}
		const float um = fmaxf(fabsf(u0), fabsf(u1)), vm = fmaxf(fabsf(v0), fabsf(v1));
		const float mag = 0.5f * (A * um * um + C * vm * vm) + fabsf(B) * um * vm;
		const float margin = 0.01f + 1e-4f * thr + 2e-5f * mag;
		if (!(qmin > thr + margin)) bits |= (1u << q);
	}
	return bits;
}

void gsro_cull_stats(const gsro_state* st, double* out)
{
	for (int i = 0; i < 16; i++) out[i] = 0;
	const int W = st->W, H = st->H;
	const int T = st->grid_x * st->grid_y;
	double o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0, o8 = 0, o9 = 0, o10 = 0, o11 = 0, o12 = 0, o13 = 0, o14 = 0, o15 = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(NT()) reduction(+ : o0, o1, o2, o3, o4, o5, o6, o7, o8, o9, o10, o11, o12, o13, o14, o15)
	for (int t = 0; t < T; t++) {
		const int tx = t % st->grid_x, ty = t / st->grid_x;
		const uint32_t rs = st->ranges[2 * t], re = st->ranges[2 * t + 1];
		const uint32_t n = re - rs;
		o0 += n;
		uint8_t* used = (uint8_t*)calloc((size_t)n + 1, 1); /* bit q: some pixel of quad q blends entry k */
		uint16_t* used16 = (uint16_t*)calloc((size_t)n + 1, 2); /* bit 4q+s: some pixel of the 4x4 block s of quad q blends entry k */
		uint32_t qmaxc[4] = {0, 0, 0, 0}; /* deepest n_contrib per quad */
		uint32_t qdone_at[4] = {0, 0, 0, 0}; /* entries the fwd quad walks before all its pixels are done */
		for (int q = 0; q < 4; q++)
			for (int l = 0; l < 64; l++) {
				int px = tx * 16 + (q & 1) * 8 + (l & 7), py = ty * 16 + (q >> 1) * 8 + (l >> 3);
				if (px >= W || py >= H) continue;
				uint32_t nc = st->n_contrib[(size_t)py * W + px];
				if (nc > qmaxc[q]) qmaxc[q] = nc;
				/* forward walk length of this pixel: until termination or end of list */
				float T_ = 1.f; uint32_t k;
				for (k = 0; k < n; k++) {
					uint32_t g = st->point_list[rs + k];
					float dx = st->means2D[2 * g] - (float)px, dy = st->means2D[2 * g + 1] - (float)py;
					const float* co = st->conic_opacity + 4 * (size_t)g;
					float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
					o5 += 1;
					if (power > 0.0f) continue;
					float alpha = fminf(0.99f, co[3] * expf(power));
					if (alpha < 1.0f / 255.0f) continue;
					float test_T = T_ * (1 - alpha);
					if (test_T < 0.0001f) { k++; break; }
					T_ = test_T;
					o4 += 1;
					used[k] |= (uint8_t)(1u << q);
					used16[k] |= (uint16_t)(1u << (4 * q + ((l >> 5) << 1) + ((l & 7) >> 2)));
					uint32_t bits = quad_keep_bits_ref(st->means2D + 2 * g, co, (float)(tx * 16), (float)(ty * 16));
					if (!((bits >> q) & 1)) o6 += 1;
				}
				if (k > qdone_at[q]) qdone_at[q] = k;
			}
		uint32_t bmax = 0;
		for (int q = 0; q < 4; q++) if (qmaxc[q] > bmax) bmax = qmaxc[q];
		o1 += bmax;
		for (uint32_t k = 0; k < n; k++) {
			uint32_t g = st->point_list[rs + k];
			uint32_t bits = quad_keep_bits_ref(st->means2D + 2 * g, st->conic_opacity + 4 * (size_t)g, (float)(tx * 16), (float)(ty * 16));
			for (int q = 0; q < 4; q++) {
				if (k < qdone_at[q]) { o7 += 1; if ((bits >> q) & 1) o2 += 1; }
				if (k < qmaxc[q] && ((bits >> q) & 1)) o3 += 1;
				if ((used[k] >> q) & 1) o8 += 1;   /* (quad, entry) visits with at least one blending pixel */
			}
			if (used[k]) o9 += 1;                   /* (tile, entry) instances with at least one blending pixel */
			/* what a 16x8 (two quads side by side) or 8x16 unit of execution would visit in the backward pass */
			{
				const uint32_t top = qmaxc[0] > qmaxc[1] ? qmaxc[0] : qmaxc[1], bot = qmaxc[2] > qmaxc[3] ? qmaxc[2] : qmaxc[3];
				const uint32_t lef = qmaxc[0] > qmaxc[2] ? qmaxc[0] : qmaxc[2], rig = qmaxc[1] > qmaxc[3] ? qmaxc[1] : qmaxc[3];
				if (k < top && (bits & 3u)) o10 += 1;
				if (k < bot && (bits & 12u)) o10 += 1;
				if (k < lef && (bits & 5u)) o11 += 1;
				if (k < rig && (bits & 10u)) o11 += 1;
			}
		}
		/* finer units of execution inside a quad: four 4x4 blocks (16 lanes each, every block walking its own entries) or
		 * two 8x4 halves: entries with a blending pixel per unit, and the longest unit of each quad (= its walk length) */
		for (int q = 0; q < 4; q++) {
			uint32_t c4[4] = {0, 0, 0, 0}, c2[2] = {0, 0}, cq = 0;
			for (uint32_t k = 0; k < n; k++) {
				const uint32_t b = (used16[k] >> (4 * q)) & 15u;
				for (int s_ = 0; s_ < 4; s_++) c4[s_] += (b >> s_) & 1u;
				c2[0] += (b & 3u) != 0; c2[1] += (b & 12u) != 0;
				cq += b != 0;
			}
			o12 += c4[0] + c4[1] + c4[2] + c4[3];
			uint32_t m4 = c4[0]; for (int s_ = 1; s_ < 4; s_++) if (c4[s_] > m4) m4 = c4[s_];
			o13 += m4;
			o14 += c2[0] > c2[1] ? c2[0] : c2[1];
			o15 += c2[0] + c2[1];
		}
		free(used);
		free(used16);
	}
	out[12] = o12; out[13] = o13; out[14] = o14; out[15] = o15;
	out[8] = o8; out[9] = o9; out[10] = o10; out[11] = o11;
	out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = o4; out[5] = o5; out[6] = o6; out[7] = o7;
}


/* ---------------- Photo-SLAM point kernels ---------------- */
/* transform_points, src/operate_points.cu:38-50 */
void gsro_transform_points(int P, const float* pts, const float* m, float* out)
{
	for (int i = 0; i < P; i++) {
		f3 p = pt(pts, (uint32_t)i);
		f3 t = transformPoint4x3(p, m);
		out[3 * (size_t)i] = t.x; out[3 * (size_t)i + 1] = t.y; out[3 * (size_t)i + 2] = t.z;
	}
}
/* scale_and_transform_points, src/operate_points.cu:52-71 with the helpers of
 * cuda_rasterizer/operate_points.h:55-179.  reference_rot_layout: see gsr.h. */
void gsro_scale_transform_points(int P, float scale, const float* pts, const float* rots, const float* m,
                                 const uint8_t* mask, float* out_pts, float* out_rots, int reference_rot_layout)
{
	for (int idx = 0; idx < P; idx++) {
		if (!mask[idx]) continue;
		f3 p = pt(pts, (uint32_t)idx);
		p.x *= scale; p.y *= scale; p.z *= scale;
		f3 t3 = transformPoint4x3(p, m);
		out_pts[3 * (size_t)idx] = t3.x; out_pts[3 * (size_t)idx + 1] = t3.y; out_pts[3 * (size_t)idx + 2] = t3.z;
		/* q_orig = {x = q[1], y = q[2], z = q[3], w = q[0]}, operate_points.h:77-80 */
		float qx = rots[4 * (size_t)idx + 1], qy = rots[4 * (size_t)idx + 2], qz = rots[4 * (size_t)idx + 3], qw = rots[4 * (size_t)idx];
		float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
		float twx = tx * qw, twy = ty * qw, twz = tz * qw;
		float txx = tx * qx, txy = ty * qx, txz = tz * qx;
		float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
		float R00 = 1.0f - (tyy + tzz), R01 = txy - twz, R02 = txz + twy;
		float R10 = txy + twz, R11 = 1.0f - (txx + tzz), R12 = tyz - twx;
		float R20 = txz - twy, R21 = tyz + twx, R22 = 1.0f - (txx + tyy);
		float R[3][3];
		R[0][0] = m[0] * R00 + m[4] * R10 + m[8] * R20;
		R[0][1] = m[0] * R01 + m[4] * R11 + m[8] * R21;
		R[0][2] = m[0] * R02 + m[4] * R12 + m[8] * R22;
		R[1][0] = m[1] * R00 + m[5] * R10 + m[9] * R20;
		R[1][1] = m[1] * R01 + m[5] * R11 + m[9] * R21;
		R[1][2] = m[1] * R02 + m[5] * R12 + m[9] * R22;
		R[2][0] = m[2] * R00 + m[6] * R10 + m[10] * R20;
		R[2][1] = m[2] * R01 + m[6] * R11 + m[10] * R21;
		R[2][2] = m[2] * R02 + m[6] * R12 + m[10] * R22;
		float ow, ox, oy, oz;
		float t = R[0][0] + R[1][1] + R[2][2];
		if (t > 0.0f) {
			t = sqrtf(t + 1.0f);
			ow = 0.5f * t;
			t = 0.5f / t;
			ox = (R[2][1] - R[1][2]) * t;
			oy = (R[0][2] - R[2][0]) * t;
			oz = (R[1][0] - R[0][1]) * t;
		} else {
			int i = 0;
			if (R[1][1] > R[0][0]) i = 1;
			if (R[2][2] > R[i][i]) i = 2;
			int j = (i + 1) % 3, k = (j + 1) % 3;
			t = sqrtf(R[i][i] - R[j][j] - R[k][k] + 1.0f);
			float xyz[3];
			xyz[i] = 0.5f * t;
			t = 0.5f / t;
			ow = (R[k][j] - R[j][k]) * t;
			xyz[j] = (R[j][i] + R[i][j]) * t;
			xyz[k] = (R[k][i] + R[i][k]) * t;
			ox = xyz[0]; oy = xyz[1]; oz = xyz[2];
		}
		float* r = out_rots + 4 * (size_t)idx;
		r[0] = ow; r[1] = ox;
		if (reference_rot_layout) { r[2] = oy; r[2] = oz; } /* operate_points.h:175-178 verbatim: +2 twice, +3 never */
		else { r[2] = oy; r[3] = oz; }
	}
}
/* reproject_depths_pinhole, src/stereo_vision.cu:39-61 */
void gsro_reproject_depth_pinhole(int P, int width, float fx, float fy, float cx, float cy, const float* depths,
                                  const uint8_t* mask, float* points)
{
	for (int idx = 0; idx < P; idx++) {
		if (!mask[idx]) continue;
		int v = idx / width, u = idx - v * width;
		float depth = depths[idx];
		points[3 * (size_t)idx] = (u - cx) * depth / fx;
		points[3 * (size_t)idx + 1] = (v - cy) * depth / fy;
		points[3 * (size_t)idx + 2] = depth;
	}
}
/* search_neighborhood_to_estimate_depth_and_reproject_pinhole, src/stereo_vision.cu:63-136 */
void gsro_neighborhood_depth_pinhole(int N, int width, float fx, float fy, float cx, float cy, float max_pixel_dist,
                                     const float* pixels, const uint8_t* has3D, const float* p3d, const float* colors,
                                     float* out_p, float* out_c)
{
	for (int idx = 0; idx < N; idx++) {
		float u = pixels[2 * idx], v = pixels[2 * idx + 1];
		int ptidx = idx * 3;
		int px = (int)(v * width + u);
		if (has3D[idx]) {
			out_p[ptidx] = p3d[ptidx]; out_p[ptidx + 1] = p3d[ptidx + 1]; out_p[ptidx + 2] = p3d[ptidx + 2];
			out_c[ptidx] = colors[px]; out_c[ptidx + 1] = colors[px + 1]; out_c[ptidx + 2] = colors[px + 2];
			continue;
		}
		float min_dist = FLT_MAX, depth = -1.0f;
		for (int i = 0; i < N; ++i) {
			if (!has3D[i] || i == idx) continue;
			float du = u - pixels[2 * i], dv = v - pixels[2 * i + 1];
			float dist = du * du + dv * dv;
			if (dist > max_pixel_dist || dist >= min_dist) continue;
			min_dist = dist;
			depth = p3d[i * 3 + 2];
		}
		if (depth > 0.0f) {
			int ui = (int)u, vi = (int)v;
			out_p[ptidx] = (ui - cx) * depth / fx;
			out_p[ptidx + 1] = (vi - cy) * depth / fy;
			out_p[ptidx + 2] = depth;
			out_c[ptidx] = colors[px]; out_c[ptidx + 1] = colors[px + 1]; out_c[ptidx + 2] = colors[px + 2];
		} else {
			out_p[ptidx + 2] = -1.0f;
		}
	}
}
