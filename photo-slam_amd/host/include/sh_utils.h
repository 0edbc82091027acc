// sh_utils.h -- spherical-harmonics helpers of the LibTorch host: what include/sh_utils.h:64-146 of the reference provides
// (eval_sh for degrees 0..3, RGB2SH, SH2RGB), written as a basis matrix times the coefficients instead of the reference's
// term-by-term expression: colour[n][c] = sum_k basis_k(dir[n]) * sh[n][c][k].  Constants and basis functions are those of
// computeColorFromSH (cuda_rasterizer/forward.cu:20-71, auxiliary.h:22-39), which the in-kernel evaluation uses too.
#pragma once
#include <torch/torch.h>

#include <vector>

namespace sh_utils {

constexpr double C0 = 0.28209479177387814;

// [N, (deg+1)^2] real SH basis at the unit directions dirs [N,3]
inline torch::Tensor basis(int deg, const torch::Tensor& dirs)
{
	TORCH_CHECK(deg >= 0 && deg <= 3, "eval_sh: degrees 0..3 are supported");
	const double C1 = 0.4886025119029199;
	const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
	const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
	                      -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
	auto x = dirs.select(-1, 0), y = dirs.select(-1, 1), z = dirs.select(-1, 2);
	std::vector<torch::Tensor> b = {torch::full_like(x, C0)};
	if (deg > 0) {
		b.push_back(-C1 * y);
		b.push_back(C1 * z);
		b.push_back(-C1 * x);
	}
	if (deg > 1) {
		auto xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
		b.push_back(C2[0] * xy);
		b.push_back(C2[1] * yz);
		b.push_back(C2[2] * (2.0 * zz - xx - yy));
		b.push_back(C2[3] * xz);
		b.push_back(C2[4] * (xx - yy));
		if (deg > 2) {
			b.push_back(C3[0] * y * (3.0 * xx - yy));
			b.push_back(C3[1] * xy * z);
			b.push_back(C3[2] * y * (4.0 * zz - xx - yy));
			b.push_back(C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy));
			b.push_back(C3[4] * x * (4.0 * zz - xx - yy));
			b.push_back(C3[5] * z * (xx - yy));
			b.push_back(C3[6] * x * (xx - 3.0 * yy));
		}
	}
	return torch::stack(b, -1);
}

// sh [N, C, >= (deg+1)^2], dirs [N,3] (unit) -> [N, C]      (include/sh_utils.h:64-136)
inline torch::Tensor eval_sh(int deg, const torch::Tensor& sh, const torch::Tensor& dirs)
{
	const int64_t K = static_cast<int64_t>(deg + 1) * (deg + 1);
	TORCH_CHECK(sh.size(-1) >= K, "eval_sh: not enough coefficients for the degree");
	return (sh.slice(-1, 0, K) * basis(deg, dirs).unsqueeze(-2)).sum(-1);
}

inline torch::Tensor RGB2SH(const torch::Tensor& rgb) { return (rgb - 0.5) / C0; }   // :138-141
inline torch::Tensor SH2RGB(const torch::Tensor& sh) { return sh * C0 + 0.5; }       // :143-146

}  // namespace sh_utils
