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

// ---- the calls the kernels do not cover (another window, no size average, a batch, a target with a gradient, host tensors).
// Same names, signatures and values as the reference's header (the drop-in contract: include/loss_utils.h:33-108), computed this
// library's way: the SSIM window is the outer product of one normalised 1-D Gaussian, so every local statistic is TWO 1-D
// convolutions (11 + 11 taps per pixel instead of 121), and the five statistics of a pair of images -- means, second moments, the
// mixed moment -- ride in ONE grouped convolution pair over a 5C-channel stack.

torch::Tensor psnr(torch::Tensor &img1, torch::Tensor &img2)
{
	// 10 log10(1 / mse), mse over the whole tensor
	return -10.0f * torch::log10(torch::mse_loss(img1, img2));
}

torch::Tensor psnr_gaussian_splatting(torch::Tensor &img1, torch::Tensor &img2)
{
	// the Inria evaluation's form: per image of the batch 20 log10(1 / rmse), then the mean over the batch
	const auto per_image = (img1 - img2).square().flatten(1).mean(1, /*keepdim=*/true);
	return (-10.0f * torch::log10(per_image)).mean();
}

torch::Tensor gaussian(int window_size, float sigma, torch::DeviceType device_type)
{
	// taps exp(-(x - window_size / 2)^2 / (2 sigma^2)), x = 0 .. window_size - 1 (integer centre), normalised to sum 1
	const auto x = torch::arange(window_size, torch::TensorOptions().dtype(torch::kFloat32).device(device_type)) - static_cast<float>(window_size / 2);
	const auto w = torch::exp(x.square() * (-1.0f / (2.0f * sigma * sigma)));
	return w / w.sum();
}

torch::autograd::Variable create_window(int window_size, int64_t channel, torch::DeviceType device_type)
{
	// [channel, 1, k, k]: the same 2-D window for every channel (a grouped convolution's weight)
	const auto taps = gaussian(window_size, 1.5f, device_type);
	return torch::outer(taps, taps).to(torch::kFloat).expand({channel, 1, window_size, window_size}).contiguous();
}

namespace {
// local window sums of every channel of x [N, C, H, W] (zero padding, as conv2d with padding k / 2): separable when the window is
// the outer product of its marginals (any window create_window makes), the plain grouped 2-D convolution otherwise
torch::Tensor window_filter(const torch::Tensor& x, const torch::Tensor& window2d, int window_size)
{
	namespace F = torch::nn::functional;
	const int64_t C = x.size(1);
	const int pad = window_size / 2;
	const auto k2 = window2d.select(0, 0).select(0, 0);                     // [k, k] (the same for every channel)
	const auto col = k2.sum(1), row = k2.sum(0);                            // marginals: vertical and horizontal taps
	const auto total = k2.sum();
	const bool separable = torch::allclose(torch::outer(col, row), k2 * total, 1e-6, 1e-9);
	if (!separable)
		return F::conv2d(x, k2.reshape({1, 1, window_size, window_size}).expand({C, 1, window_size, window_size}), F::Conv2dFuncOptions().padding(pad).groups(C));
	const auto horizontal = (row / total).reshape({1, 1, 1, window_size}).expand({C, 1, 1, window_size});
	const auto vertical = col.reshape({1, 1, window_size, 1}).expand({C, 1, window_size, 1});
	const auto h = F::conv2d(x, horizontal, F::Conv2dFuncOptions().padding({0, pad}).groups(C));
	return F::conv2d(h, vertical, F::Conv2dFuncOptions().padding({pad, 0}).groups(C));
}
}  // namespace

torch::Tensor _ssim(torch::Tensor &img1, torch::Tensor &img2, torch::autograd::Variable &window, int window_size, int64_t channel,
                    bool size_average)
{
	const bool batched = img1.dim() == 4;
	const auto a = batched ? img1 : img1.unsqueeze(0), b = batched ? img2 : img2.unsqueeze(0);
	// one filter pass over [a | b | a a | b b | a b]
	const auto stats = window_filter(torch::cat({a, b, a * a, b * b, a * b}, 1), window, window_size).split(channel, 1);
	const auto& mean_a = stats[0];
	const auto& mean_b = stats[1];
	const auto cross = mean_a * mean_b, energy = mean_a.square() + mean_b.square();
	const auto spread = stats[2] + stats[3] - energy;      // var(a) + var(b)
	const auto covariance = stats[4] - cross;
	const double c1 = 1e-4, c2 = 9e-4;                      // (0.01 L)^2, (0.03 L)^2 with a dynamic range L of 1
	auto map = ((2 * cross + c1) * (2 * covariance + c2)) / ((energy + c1) * (spread + c2));
	if (!batched) map = map.squeeze(0);
	if (size_average) return map.mean();
	return map.mean(1).mean(1).mean(1);   // (the reference's reduction order: dims 1, 1, 1 of what is left)
}

torch::Tensor ssim(torch::Tensor &img1, torch::Tensor &img2, torch::DeviceType device_type, int window_size, bool size_average)
{
	// the train step's call (11 x 11 window, sigma 1.5, mean over the map): 1 - [(1 - lambda) L1 + lambda (1 - SSIM)] at lambda = 1
	if (window_size == 11 && size_average && fused_applies(img1, img2))
		return 1.0 - FusedL1SSIMFunction::apply(img1, img2, torch::empty({0}, img2.options()), 1.0, false);
	const int64_t channels = img1.size(-3);
	torch::autograd::Variable window = create_window(window_size, channels, img1.device().type()).to(img1.scalar_type());
	return _ssim(img1, img2, window, window_size, channels, size_average);
}

}
