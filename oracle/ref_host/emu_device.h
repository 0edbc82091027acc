/*
 * oracle/ref_host/emu_device.h -- force-included (g++ -include) in front of the reference's host sources when they are
 * compiled for THIS GPU-less container against the emulator build of the kernels.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference hard-codes the device: torch::kCUDA (src/gaussian_renderer.cpp:42, src/gaussian_trainer.cpp:42,62,
 * include/general_utils.h:45, include/loss_utils.h:52,68,113), `.cuda()` (src/gaussian_trainer.cpp:81) and
 * torch::cuda::synchronize() (:86).  After LibTorch's own headers have been read, the three spellings are mapped to the host
 * so that the unchanged sources run on CPU tensors.  The HIP build does NOT see this file: there the sources compile with no
 * prefix at all (on a ROCm LibTorch kCUDA is the HIP device).
 */
#pragma once
#include <torch/torch.h>

namespace torch { namespace cpu {
inline void synchronize(int64_t = -1) {}
}}  // namespace torch::cpu

#define REF_DEVICE torch::kCPU
#define REF_HOST_NO_CACHING_ALLOCATOR 1
namespace c10 { namespace cpu { namespace CUDACachingAllocator {
inline void emptyCache() {}   /* src/gaussian_model.cpp:814; the host has no caching allocator */
}}}
#define kCUDA kCPU
#define cuda cpu
