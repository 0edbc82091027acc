/* A plain-C consumer of include/gsr.h: no C++, no torch -- malloc'd buffers, C callbacks for the scratch allocations.
 * Links against any build of the C-ABI (the test-suite uses the host emulator build; on a GPU box the same source links
 * libgsr_hip.so with hipMalloc'd buffers).  Renders a handful of Gaussians, runs the backward pass, prints checksums. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsr.h"

/* Where the library's pointers live: host memory for the emulator build, HBM (hipMalloc through the C runtime API of
 * HIP -- still no C++) for libgsr_hip.so: cc -DGSR_CONSUMER_HIP -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -lamdhip64 */
#ifdef GSR_CONSUMER_HIP
#include <hip/hip_runtime_api.h>
static void* dev_alloc(size_t bytes)
{
	void* p = NULL;
	if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { printf("hipMalloc failed\n"); exit(2); }
	return p;
}
static void dev_free(void* p) { if (p) (void)hipFree(p); }
static void* to_dev(const void* host, size_t bytes)
{
	void* p = dev_alloc(bytes);
	if (hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) { printf("hipMemcpy H2D failed\n"); exit(2); }
	return p;
}
static void to_host(void* host, const void* dev, size_t bytes)
{
	if (hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { printf("hipMemcpy D2H failed\n"); exit(2); }
}
#else
static void* dev_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
static void dev_free(void* p) { free(p); }
static void* to_dev(const void* host, size_t bytes) { void* p = dev_alloc(bytes); memcpy(p, host, bytes); return p; }
static void to_host(void* host, const void* dev, size_t bytes) { memcpy(host, dev, bytes); }
#endif

typedef struct { char* p; size_t n; } buf_t;
static char* grow(void* ctx, size_t bytes)
{
	buf_t* b = (buf_t*)ctx;
	if (bytes > b->n) { dev_free(b->p); b->p = (char*)dev_alloc(bytes); b->n = bytes; }
	return b->p;
}

int main(void)
{
	enum { P = 5, W = 40, H = 24, M = 16 };
	float means[P * 3], sh[P * M * 3], opac[P], scales[P * 3], rots[P * 4];
	memset(sh, 0, sizeof sh);
	for (int i = 0; i < P; i++) {
		means[3 * i] = -0.6f + 0.3f * (float)i; means[3 * i + 1] = 0.1f * (float)(i - 2); means[3 * i + 2] = 2.0f + 0.25f * (float)i;
		opac[i] = 0.5f + 0.08f * (float)i;
		scales[3 * i] = 0.12f; scales[3 * i + 1] = 0.08f + 0.01f * (float)i; scales[3 * i + 2] = 0.1f;
		rots[4 * i] = 1.f; rots[4 * i + 1] = 0.1f * (float)i; rots[4 * i + 2] = 0.f; rots[4 * i + 3] = 0.05f;
		for (int c = 0; c < 3; c++) sh[(i * M) * 3 + c] = 0.3f + 0.2f * (float)((i + c) % 3);   /* DC term only */
	}
	/* camera at the origin looking down +z; matrices stored as the reference does (transposed, glm column-major) */
	const float tanfov = 0.6f, zn = 0.01f, zf = 100.f;
	float view[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	float proj[16];
	memset(proj, 0, sizeof proj);
	proj[0] = 1.f / tanfov; proj[5] = 1.f / tanfov; proj[10] = zf / (zf - zn); proj[11] = 1.f; proj[14] = -(zf * zn) / (zf - zn);
	float campos[3] = {0, 0, 0}, bg[3] = {0.1f, 0.2f, 0.3f};
	float* out = (float*)malloc(sizeof(float) * 3 * W * H);
	int radii[P];
	buf_t geom = {0, 0}, binning = {0, 0}, img = {0, 0};
	float *d_means = to_dev(means, sizeof means), *d_sh = to_dev(sh, sizeof sh), *d_opac = to_dev(opac, sizeof opac);
	float *d_scales = to_dev(scales, sizeof scales), *d_rots = to_dev(rots, sizeof rots), *d_view = to_dev(view, sizeof view);
	float *d_proj = to_dev(proj, sizeof proj), *d_campos = to_dev(campos, sizeof campos), *d_bg = to_dev(bg, sizeof bg);
	float* d_out = dev_alloc(sizeof(float) * 3 * W * H);
	int* d_radii = dev_alloc(sizeof radii);

	gsr_forward_args f;
	memset(&f, 0, sizeof f);
	f.P = P; f.D = 0; f.M = M; f.background = d_bg; f.width = W; f.height = H;
	f.means3D = d_means; f.shs = d_sh; f.opacities = d_opac; f.scales = d_scales; f.scale_modifier = 1.f; f.rotations = d_rots;
	f.viewmatrix = d_view; f.projmatrix = d_proj; f.cam_pos = d_campos; f.tan_fovx = tanfov; f.tan_fovy = tanfov;
	f.out_color = d_out; f.radii = d_radii;
	int R = -1;
	int st = gsr_forward(&f, grow, &geom, grow, &binning, grow, &img, NULL, &R);
	if (st != GSR_OK) { printf("forward failed: %s\n", gsr_strerror(st)); return 1; }
	to_host(out, d_out, sizeof(float) * 3 * W * H);
	to_host(radii, d_radii, sizeof radii);

	float* dpix = (float*)malloc(sizeof(float) * 3 * W * H);
	for (int i = 0; i < 3 * W * H; i++) dpix[i] = 1.f;
	float* d_dpix = to_dev(dpix, sizeof(float) * 3 * W * H);
	float dop[P], d3[P * 3], dsh[P * M * 3], dsc[P * 3], drot[P * 4];
	float *g_d2 = dev_alloc(sizeof(float) * P * 3), *g_dcon = dev_alloc(sizeof(float) * P * 4), *g_dop = dev_alloc(sizeof dop);
	float *g_dcol = dev_alloc(sizeof(float) * P * 3), *g_d3 = dev_alloc(sizeof d3), *g_dcov = dev_alloc(sizeof(float) * P * 6);
	float *g_dsh = dev_alloc(sizeof dsh), *g_dsc = dev_alloc(sizeof dsc), *g_drot = dev_alloc(sizeof drot);
	gsr_backward_args b;
	memset(&b, 0, sizeof b);
	b.P = P; b.D = 0; b.M = M; b.R = R; b.background = d_bg; b.width = W; b.height = H;
	b.means3D = d_means; b.shs = d_sh; b.scales = d_scales; b.scale_modifier = 1.f; b.rotations = d_rots;
	b.viewmatrix = d_view; b.projmatrix = d_proj; b.campos = d_campos; b.tan_fovx = tanfov; b.tan_fovy = tanfov; b.radii = d_radii;
	b.geom_buffer = geom.p; b.binning_buffer = binning.p; b.image_buffer = img.p; b.dL_dpix = d_dpix;
	b.dL_dmean2D = g_d2; b.dL_dconic = g_dcon; b.dL_dopacity = g_dop; b.dL_dcolor = g_dcol; b.dL_dmean3D = g_d3; b.dL_dcov3D = g_dcov;
	b.dL_dsh = g_dsh; b.dL_dscale = g_dsc; b.dL_drot = g_drot;
	st = gsr_backward(&b, NULL);
	if (st != GSR_OK) { printf("backward failed: %s\n", gsr_strerror(st)); return 1; }
	to_host(dop, g_dop, sizeof dop); to_host(d3, g_d3, sizeof d3); to_host(dsh, g_dsh, sizeof dsh);
	to_host(dsc, g_dsc, sizeof dsc); to_host(drot, g_drot, sizeof drot);

	double sum = 0, gsum = 0;
	for (int i = 0; i < 3 * W * H; i++) sum += out[i];
	for (int i = 0; i < P; i++) gsum += fabs(dop[i]) + fabs(d3[3 * i]) + fabs(dsc[3 * i]) + fabs(drot[4 * i + 1]) + fabs(dsh[i * M * 3]);
	int vis = 0;
	for (int i = 0; i < P; i++) vis += radii[i] > 0;
	printf("backend=%s R=%d visible=%d image_sum=%.6f grad_sum=%.6f\n", gsr_backend(), R, vis, sum, gsum);
	free(out); free(dpix); dev_free(geom.p); dev_free(binning.p); dev_free(img.p);
	{
		void* d[] = {d_means, d_sh, d_opac, d_scales, d_rots, d_view, d_proj, d_campos, d_bg, d_out, d_radii, d_dpix,
		             g_d2, g_dcon, g_dop, g_dcol, g_d3, g_dcov, g_dsh, g_dsc, g_drot};
		for (size_t i = 0; i < sizeof d / sizeof d[0]; i++) dev_free(d[i]);
	}
	return 0;
}
