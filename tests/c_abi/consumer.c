/* A plain-C consumer of include/gsr.h: no C++, no torch -- malloc'd buffers, C callbacks for the scratch allocations.
 * Links against any build of the C-ABI (the test-suite uses the host emulator build; on a GPU box the same source links
 * libgsr_hip.so with hipMalloc'd buffers).  Renders a handful of Gaussians, runs the backward pass, prints checksums. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsr.h"

typedef struct { char* p; size_t n; } buf_t;
static char* grow(void* ctx, size_t bytes)
{
	buf_t* b = (buf_t*)ctx;
	if (bytes > b->n) { free(b->p); b->p = (char*)malloc(bytes ? bytes : 1); b->n = bytes; }
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

	gsr_forward_args f;
	memset(&f, 0, sizeof f);
	f.P = P; f.D = 0; f.M = M; f.background = bg; f.width = W; f.height = H;
	f.means3D = means; f.shs = sh; f.opacities = opac; f.scales = scales; f.scale_modifier = 1.f; f.rotations = rots;
	f.viewmatrix = view; f.projmatrix = proj; f.cam_pos = campos; f.tan_fovx = tanfov; f.tan_fovy = tanfov;
	f.out_color = out; f.radii = radii;
	int R = -1;
	int st = gsr_forward(&f, grow, &geom, grow, &binning, grow, &img, NULL, &R);
	if (st != GSR_OK) { printf("forward failed: %s\n", gsr_strerror(st)); return 1; }

	float* dpix = (float*)malloc(sizeof(float) * 3 * W * H);
	for (int i = 0; i < 3 * W * H; i++) dpix[i] = 1.f;
	float d2[P * 3], dcon[P * 4], dop[P], dcol[P * 3], d3[P * 3], dcov[P * 6], dsh[P * M * 3], dsc[P * 3], drot[P * 4];
	gsr_backward_args b;
	memset(&b, 0, sizeof b);
	b.P = P; b.D = 0; b.M = M; b.R = R; b.background = bg; b.width = W; b.height = H;
	b.means3D = means; b.shs = sh; b.scales = scales; b.scale_modifier = 1.f; b.rotations = rots;
	b.viewmatrix = view; b.projmatrix = proj; b.campos = campos; b.tan_fovx = tanfov; b.tan_fovy = tanfov; b.radii = radii;
	b.geom_buffer = geom.p; b.binning_buffer = binning.p; b.image_buffer = img.p; b.dL_dpix = dpix;
	b.dL_dmean2D = d2; b.dL_dconic = dcon; b.dL_dopacity = dop; b.dL_dcolor = dcol; b.dL_dmean3D = d3; b.dL_dcov3D = dcov;
	b.dL_dsh = dsh; b.dL_dscale = dsc; b.dL_drot = drot;
	st = gsr_backward(&b, NULL);
	if (st != GSR_OK) { printf("backward failed: %s\n", gsr_strerror(st)); return 1; }

	double sum = 0, gsum = 0;
	for (int i = 0; i < 3 * W * H; i++) sum += out[i];
	for (int i = 0; i < P; i++) gsum += fabs(dop[i]) + fabs(d3[3 * i]) + fabs(dsc[3 * i]) + fabs(drot[4 * i + 1]) + fabs(dsh[i * M * 3]);
	int vis = 0;
	for (int i = 0; i < P; i++) vis += radii[i] > 0;
	printf("backend=%s R=%d visible=%d image_sum=%.6f grad_sum=%.6f\n", gsr_backend(), R, vis, sum, gsum);
	free(out); free(dpix); free(geom.p); free(binning.p); free(img.p);
	return 0;
}
