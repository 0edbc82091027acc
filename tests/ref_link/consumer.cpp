// A translation unit compiled against the REFERENCE's own declarations -- include/rasterize_points.h as it is and
// include/gaussian_rasterizer.h minus its `#include "gaussian_model.h"` line (tests/test_reference_link.py generates that
// copy into a temporary directory; the full model header pulls in Sophus / OpenCV / ORB-SLAM3) -- and LINKED against this
// repository's libphotoslam_host*.so: the reference's mangled symbols (RasterizeGaussiansCUDA,
// RasterizeGaussiansBackwardCUDA, markVisible, GaussianRasterizerFunction::forward/backward, GaussianRasterizer::forward,
// ::markVisibleGaussians) and the reference's object layouts (GaussianRasterizationSettings, GaussianRasterizer) must be
// the library's.  Same five Gaussians and the same checksum line as tests/c_abi/consumer.c.
#include <torch/torch.h>

#include <cmath>
#include <cstdio>

#include "rasterize_points.h"
#include "gaussian_rasterizer.h"

int main(int argc, char** argv)
{
	const bool on_device = argc > 1 && std::string(argv[1]) == "cuda";
	const auto dev = on_device ? torch::Device(torch::kCUDA, 0) : torch::Device(torch::kCPU);
	enum { P = 5, W = 40, H = 24, M = 16 };
	auto means = torch::zeros({P, 3}), sh = torch::zeros({P, M, 3}), opac = torch::zeros({P, 1}), scales = torch::zeros({P, 3}),
	     rots = torch::zeros({P, 4});
	for (int i = 0; i < P; i++) {
		means[i][0] = -0.6f + 0.3f * (float)i; means[i][1] = 0.1f * (float)(i - 2); means[i][2] = 2.0f + 0.25f * (float)i;
		opac[i][0] = 0.5f + 0.08f * (float)i;
		scales[i][0] = 0.12f; scales[i][1] = 0.08f + 0.01f * (float)i; scales[i][2] = 0.1f;
		rots[i][0] = 1.f; rots[i][1] = 0.1f * (float)i; rots[i][3] = 0.05f;
		for (int c = 0; c < 3; c++) sh[i][0][c] = 0.3f + 0.2f * (float)((i + c) % 3);
	}
	const float tanfov = 0.6f, zn = 0.01f, zf = 100.f;
	auto view = torch::eye(4), proj = torch::zeros({4, 4});
	proj[0][0] = 1.f / tanfov; proj[1][1] = 1.f / tanfov; proj[2][2] = zf / (zf - zn); proj[2][3] = 1.f; proj[3][2] = -(zf * zn) / (zf - zn);
	auto campos = torch::zeros({3}), bg = torch::tensor({0.1f, 0.2f, 0.3f});
	auto leaf = [&](torch::Tensor t) { return t.to(dev).set_requires_grad(true); };
	means = leaf(means); sh = leaf(sh); opac = leaf(opac); scales = leaf(scales); rots = leaf(rots);
	view = view.to(dev); proj = proj.to(dev); campos = campos.to(dev); bg = bg.to(dev);
	auto empty = torch::empty({0}, torch::TensorOptions().device(dev));

	// L1: the free functions
	auto r = RasterizeGaussiansCUDA(bg, means.detach(), empty, opac.detach(), scales.detach(), rots.detach(), 1.0f, empty, view,
	                                proj, tanfov, tanfov, H, W, sh.detach(), 0, campos, false);
	const int R = std::get<0>(r);
	auto radii = std::get<2>(r);
	auto g = RasterizeGaussiansBackwardCUDA(bg, means.detach(), radii, empty, scales.detach(), rots.detach(), 1.0f, empty, view,
	                                        proj, tanfov, tanfov, torch::ones({3, H, W}, torch::TensorOptions().device(dev)),
	                                        sh.detach(), 0, campos, std::get<3>(r), R, std::get<4>(r), std::get<5>(r));
	auto md = means.detach();
	auto present = markVisible(md, view, proj);

	// L2: the classes, objects constructed HERE with the reference's layout
	GaussianRasterizationSettings settings(H, W, tanfov, tanfov, bg, 1.0f, view, proj, 0, campos, false);
	GaussianRasterizer rasterizer(settings);
	auto means2D = torch::zeros_like(means, torch::TensorOptions().requires_grad(true));
	auto out = rasterizer.forward(means, means2D, opac, true, false, true, true, false, sh, empty, scales, rots, empty);
	auto color = std::get<0>(out);
	color.sum().backward();
	auto vis = rasterizer.markVisibleGaussians(md);

	const double l1 = (color.detach() - std::get<1>(r)).abs().max().item<double>();
	const double gd = (opac.grad() - std::get<2>(g)).abs().max().item<double>() + (means.grad() - std::get<3>(g)).abs().max().item<double>();
	const double gsum = opac.grad().abs().sum().item<double>() + means.grad().select(1, 0).abs().sum().item<double>() +
	                    scales.grad().select(1, 0).abs().sum().item<double>() + rots.grad().select(1, 1).abs().sum().item<double>() +
	                    sh.grad().select(1, 0).select(1, 0).abs().sum().item<double>();
	std::printf("backend=%s R=%d visible=%d image_sum=%.6f grad_sum=%.6f class_vs_free=%.3g present=%d vis=%d\n",
	            on_device ? "hip-gfx950" : "emu-wave64", R, (int)(radii > 0).sum().item<int64_t>(),
	            color.detach().to(torch::kFloat64).sum().item<double>(), gsum, l1 + gd, (int)present.sum().item<int64_t>(),
	            (int)vis.sum().item<int64_t>());
	bool threw = false;
	try {
		rasterizer.forward(means, means2D, opac, false, false, true, true, false, empty, empty, scales, rots, empty);
	} catch (const std::runtime_error& e) {
		threw = std::string(e.what()).find("excatly one of either SHs or precomputed colors") != std::string::npos;
	}
	std::printf("reference_exception_text=%d\n", (int)threw);
	return 0;
}
