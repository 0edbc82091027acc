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
}  // namespace

TORCH_LIBRARY(photoslam_reference, m)
{
	m.def("l1_loss", &ref_l1_loss);
	m.def("ssim", &ref_ssim);
	m.def("psnr", &ref_psnr);
}
