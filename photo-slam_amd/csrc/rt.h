// rt.h -- runtime glue shared by every translation unit of libgsr_hip.so.
//
// The product build includes <hip/hip_runtime.h> and launches on a hipStream_t.
// tests/emu/ compiles the very same sources with -DGSR_EMU against a wave64
// lock-step emulator (tests/emu/hip_emu.h) so that kernel *logic* (indexing,
// barriers, ballots, sort stability, ...) is unit-tested on CPU-only machines; that
// build is test infrastructure and is never loaded by the product package.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef GSR_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include "../../include/gsr.h"

namespace gsr {

// Thread-local record of the last failing HIP call (exposed by gsr_last_hip_error()).
void set_last_hip_error(int err, const char* what);

#define GSR_HIP(call)                                              \
	do {                                                           \
		hipError_t e_ = (call);                                    \
		if (e_ != hipSuccess) {                                    \
			::gsr::set_last_hip_error((int)e_, #call);             \
			return GSR_ERR_HIP;                                    \
		}                                                          \
	} while (0)

#ifdef GSR_EMU
#define GSR_LAUNCH(kernel, grid, block, stream, ...) \
	::hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
#else
#define GSR_LAUNCH(kernel, grid, block, stream, ...) \
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, (stream), __VA_ARGS__)
#endif

#define GSR_CHECK_LAUNCH() GSR_HIP(hipGetLastError())

// Register-allocation target: at least `lo` waves per SIMD (the allocator spills rather than exceed 512/lo VGPRs).
#ifdef GSR_EMU
#define GSR_WAVES_PER_EU(lo, hi)
#else
#define GSR_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

// Carve typed arrays out of one caller-owned byte chunk, 128-byte aligned (the role of
// obtain() in cuda_rasterizer/rasterizer_impl.h:21-27; layout is our own).
struct Carver {
	char* p;
	explicit Carver(char* base) : p(base) {}
	template <typename T>
	T* take(size_t count)
	{
		uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 127) & ~uintptr_t(127);
		T* r = reinterpret_cast<T*>(a);
		p = reinterpret_cast<char*>(r + count);
		return r;
	}
	size_t used(char* base) const { return (size_t)(p - base); }
};

static inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace gsr
