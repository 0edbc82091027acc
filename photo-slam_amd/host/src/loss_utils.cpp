// loss_utils.cpp -- see host/include/loss_utils.h: the reference's loss functions (include/loss_utils.h:24-126) with l1_loss / ssim
// on the fused HIP kernels (csrc/train_ops.hip, gsr_l1_ssim_loss) behind a torch::autograd::Function.
#include "loss_utils.h"

#include <cmath>
#include <stdexcept>
#include <string>

#include "../../../include/gsr.h"

#ifndef GSR_HOST_NO_HIP
#include <c10/hip/HIPStream.h>
#endif

namespace {
void* stream_of(const torch::Tensor& t)
{
#ifndef GSR_HOST_NO_HIP
	if (t.is_cuda()) return c10::hip::getCurrentHIPStream(t.device().index()).stream();
#endif
	return nullptr;
}
void check(int status, const char* where)
{
	if (status != GSR_OK) throw std::runtime_error(std::string(where) + ": " + gsr_strerror(status));
}

// value = (1 - lambda) L1 + lambda (1 - SSIM) of (rendered * mask, gt); the gradient with respect to `rendered` is produced by
// the same pass and handed on in backward (times the upstream gradient unless the value is the root of the graph)
class FusedL1SSIMFunction : public torch::autograd::Function<FusedL1SSIMFunction> {
public:
	static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor rendered, torch::Tensor gt,
	                             torch::Tensor mask, double lambda_dssim, bool is_root)
	{
		ctx->saved_data["is_root"] = is_root;
		auto r = rendered.contiguous(), g = gt.contiguous();
		torch::Tensor m = mask.defined() && mask.numel() ? mask.contiguous() : torch::Tensor();
		const int H = static_cast<int>(r.size(-2)), W = static_cast<int>(r.size(-1));
		auto grad = torch::empty_like(r);
		auto loss = torch::empty({1}, r.options());
		auto scratch = torch::empty({static_cast<int64_t>(gsr_loss_scratch_bytes(W, H))}, r.options().dtype(torch::kByte));
		check(gsr_l1_ssim_loss(r.data_ptr<float>(), g.data_ptr<float>(), m.defined() ? m.data_ptr<float>() : nullptr, W,
		                       H, static_cast<float>(lambda_dssim), grad.data_ptr<float>(), loss.data_ptr<float>(),
		                       reinterpret_cast<char*>(scratch.data_ptr()), stream_of(r)),
		      "gsr_l1_ssim_loss");
		ctx->save_for_backward({grad});
		return loss[0];
	}
	static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list go)
	{
		auto grad = ctx->get_saved_variables()[0];
		if (ctx->saved_data["is_root"].toBool()) return {grad, torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
		return {grad * go[0], torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
	}
};

// what the kernels cover: one float32 [3,H,W] (or [1,3,H,W]) image pair on one device, the target without a gradient
bool fused_applies(const torch::Tensor& a, const torch::Tensor& b)
{
	if (!a.defined() || !b.defined() || a.sizes() != b.sizes() || a.scalar_type() != torch::kFloat32 || b.scalar_type() != torch::kFloat32 ||
	    a.device() != b.device() || b.requires_grad())
		return false;
	if (a.dim() == 4 && a.size(0) != 1) return false;
	if (a.dim() != 3 && a.dim() != 4) return false;
	if (a.size(-3) != 3) return false;
#ifndef GSR_HOST_NO_HIP
	return a.is_cuda();
#else
	return !a.is_cuda();   // the emulator build of the test-suite runs the kernels on host tensors
#endif
}
}  // namespace

namespace loss_utils
{

torch::Tensor fused_l1_ssim(torch::Tensor rendered, torch::Tensor gt, torch::Tensor mask, float lambda_dssim, bool is_root)
{
	if (!mask.defined()) mask = torch::empty({0}, gt.options());   // (autograd::Function::apply needs every tensor argument defined)
	return FusedL1SSIMFunction::apply(rendered, gt, mask, static_cast<double>(lambda_dssim), is_root);
}

torch::Tensor l1_loss(torch::Tensor &network_output, torch::Tensor &gt)
{
	if (fused_applies(network_output, gt)) return FusedL1SSIMFunction::apply(network_output, gt, torch::empty({0}, gt.options()), 0.0, false);   // (an empty mask = none)
	return torch::abs(network_output - gt).mean();
}

torch::Tensor psnr(torch::Tensor &img1, torch::Tensor &img2)
{
	auto mse = torch::pow(img1 - img2, 2).mean();
	return 10.0f * torch::log10(1.0f / mse);
}

torch::Tensor psnr_gaussian_splatting(torch::Tensor &img1, torch::Tensor &img2)
{
	auto mse = torch::pow(img1 - img2, 2).view({img1.size(0) , -1}).mean(1, /*keepdim=*/true);
	return 20.0f * torch::log10(1.0f / torch::sqrt(mse)).mean();
}

torch::Tensor gaussian(int window_size, float sigma, torch::DeviceType device_type)
{
	std::vector<float> gauss_values(window_size);
	for (int x = 0; x < window_size; ++x) {
		int temp = x - window_size / 2;
		gauss_values[x] = std::exp(-temp * temp / (2.0f * sigma * sigma));
	}
	torch::Tensor gauss = torch::tensor(gauss_values, torch::TensorOptions().device(device_type));
	return gauss / gauss.sum();
}

torch::autograd::Variable create_window(int window_size, int64_t channel, torch::DeviceType device_type)
{
	auto _1D_window = gaussian(window_size, 1.5f, device_type).unsqueeze(1);
	auto _2D_window = _1D_window.mm(_1D_window.t()).to(torch::kFloat).unsqueeze(0).unsqueeze(0);
	return torch::autograd::Variable(_2D_window.expand({channel, 1, window_size, window_size}).contiguous());
}

torch::Tensor _ssim(torch::Tensor &img1, torch::Tensor &img2, torch::autograd::Variable &window, int window_size, int64_t channel,
                    bool size_average)
{
	namespace F = torch::nn::functional;
	const auto opts = F::Conv2dFuncOptions().padding(window_size / 2).groups(channel);
	auto mu1 = F::conv2d(img1, window, opts);
	auto mu2 = F::conv2d(img2, window, opts);
	auto mu1_sq = mu1.pow(2);
	auto mu2_sq = mu2.pow(2);
	auto mu1_mu2 = mu1 * mu2;
	auto sigma1_sq = F::conv2d(img1 * img1, window, opts) - mu1_sq;
	auto sigma2_sq = F::conv2d(img2 * img2, window, opts) - mu2_sq;
	auto sigma12 = F::conv2d(img1 * img2, window, opts) - mu1_mu2;
	auto C1 = 0.01 * 0.01;
	auto C2 = 0.03 * 0.03;
	auto ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2));
	if (size_average) return ssim_map.mean();
	return ssim_map.mean(1).mean(1).mean(1);
}

torch::Tensor ssim(torch::Tensor &img1, torch::Tensor &img2, torch::DeviceType device_type, int window_size, bool size_average)
{
	// the train step's call (11 x 11 window, sigma 1.5, mean over the map): 1 - [(1 - lambda) L1 + lambda (1 - SSIM)] at lambda = 1
	if (window_size == 11 && size_average && fused_applies(img1, img2))
		return 1.0 - FusedL1SSIMFunction::apply(img1, img2, torch::empty({0}, img2.options()), 1.0, false);
	auto channel = img1.size(-3);
	auto window = create_window(window_size, channel, img1.device().type());
	window = window.type_as(img1);
	return _ssim(img1, img2, window, window_size, channel, size_average);
}

}
