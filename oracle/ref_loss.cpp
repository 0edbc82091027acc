/*
 * ref_loss.cpp -- torch ops around the REFERENCE's own header include/loss_utils.h (l1_loss :28-31, ssim :110-124, psnr :33-41),
 * compiled against LibTorch by oracle/build_ref.py into oracle/_ref/libref_loss.so.  TEST INFRASTRUCTURE ONLY: pins
 * photo-slam_amd/loss_utils.py, the torch mirror the fused HIP loss kernels are tested against.
 */
#include <torch/torch.h>
#include <torch/library.h>

#include "loss_utils.h"   /* the reference's own header */

namespace {
torch::Tensor ref_l1_loss(torch::Tensor a, torch::Tensor b) { return loss_utils::l1_loss(a, b); }
torch::Tensor ref_ssim(torch::Tensor a, torch::Tensor b) { return loss_utils::ssim(a, b, a.device().type()); }
torch::Tensor ref_psnr(torch::Tensor a, torch::Tensor b) { return loss_utils::psnr(a, b); }
/* ... and the rest of the header (:33-108), for the pin of host/src/loss_utils.cpp's functions of the same names */
torch::Tensor ref_ssim_ex(torch::Tensor a, torch::Tensor b, int64_t window_size, bool size_average)
{
	return loss_utils::ssim(a, b, a.device().type(), (int)window_size, size_average);
}
torch::Tensor ref_psnr_gs(torch::Tensor a, torch::Tensor b) { return loss_utils::psnr_gaussian_splatting(a, b); }
torch::Tensor ref_create_window(int64_t window_size, int64_t channel, torch::Tensor like)
{
	return loss_utils::create_window((int)window_size, channel, like.device().type());
}
}  // namespace

TORCH_LIBRARY(photoslam_reference, m)
{
	m.def("l1_loss", &ref_l1_loss);
	m.def("ssim", &ref_ssim);
	m.def("psnr", &ref_psnr);
	m.def("ssim_ex", &ref_ssim_ex);
	m.def("psnr_gaussian_splatting", &ref_psnr_gs);
	m.def("create_window", &ref_create_window);
}
