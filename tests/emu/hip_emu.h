// hip_emu.h -- a tiny wave64 lock-step emulator used ONLY by the CPU test-suite.
//
// tests/emu/build_emu.py compiles photo-slam_amd/csrc/*.hip with g++ -DGSR_EMU against this
// header, producing tests/emu/libgsr_emu.so: the same kernels, the same launch sequences
// and the same C-ABI, executed on the host.  It exists so that kernel *logic* (indexing,
// barrier placement, ballot/mask handling, sort stability, API validation) is exercised by
// `pytest -m "not gpu"` on machines without a GPU.  It is NOT a product back-end: the
// package photo-slam_amd/ never loads it and fails loudly when libgsr_hip.so is missing.
//
// Model: every GPU thread of a workgroup is a ucontext fiber; fibers run one at a time on
// the calling OS thread and switch only at __syncthreads() and at wave-level primitives, so
// execution is deterministic and atomics are trivially atomic.  Wave primitives are
// rendezvous points of the 64 lanes of a wave -- calling one from divergent control flow
// dead-locks the emulator and is reported, which is exactly the discipline the real
// kernels must keep.  __shared__ becomes a function-local static (one workgroup runs at a
// time).  DPP / readlane builtins are not emulated; csrc/wave64.h provides the portable
// meaning of each primitive here.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipEventDisableTiming = 2 };
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, int, hipStream_t)
{
	for (size_t r = 0; r < height; r++) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
	return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)2; return hipSuccess; }   // everything runs in order on the host
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)2; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;
#include <math.h>

namespace hipemu {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
void wave_sync();
// exchange one 64-bit word per lane; returns pointer to the wave's 64 slots (valid until the next exchange)
const uint64_t* wave_exchange(uint64_t mine);
int lane();

static inline unsigned long long wave_ballot(bool pred)
{
	const uint64_t* s = wave_exchange(pred ? 1 : 0);
	unsigned long long m = 0;
	for (int i = 0; i < 64; i++) m |= (unsigned long long)(s[i] & 1) << i;
	wave_sync();
	return m;
}
static inline void wave_fence() { wave_sync(); }
static inline uint32_t wave_shfl_u32(uint32_t v, int src)
{
	const uint64_t* s = wave_exchange(v);
	const uint32_t r = (uint32_t)s[src & 63];
	wave_sync();
	return r;
}
static inline uint32_t wave_incl_scan_u32(uint32_t v)
{
	const uint64_t* s = wave_exchange(v);
	uint32_t r = 0;
	for (int i = 0; i <= lane(); i++) r += (uint32_t)s[i];
	wave_sync();
	return r;
}
static inline uint32_t wave_sum_u32(uint32_t v)
{
	const uint64_t* s = wave_exchange(v);
	uint32_t r = 0;
	for (int i = 0; i < 64; i++) r += (uint32_t)s[i];
	wave_sync();
	return r;
}
static inline uint32_t wave_max_u32(uint32_t v)
{
	const uint64_t* s = wave_exchange(v);
	uint32_t r = 0;
	for (int i = 0; i < 64; i++) r = r > (uint32_t)s[i] ? r : (uint32_t)s[i];
	wave_sync();
	return r;
}
static inline void wave_reduce9_f32(float (&v)[9])
{
	for (int c = 0; c < 9; c++) {
		uint32_t bits;
		memcpy(&bits, &v[c], 4);
		const uint64_t* s = wave_exchange(bits);
		float acc = 0.f;
		for (int i = 0; i < 64; i++) {
			uint32_t b = (uint32_t)s[i];
			float f;
			memcpy(&f, &b, 4);
			acc += f;
		}
		wave_sync();
		v[c] = acc;  // the real primitive only guarantees lane 63
	}
}
static inline float mask_select_f32(unsigned long long mask, float if_set, float if_clear) { return ((mask >> lane()) & 1ull) ? if_set : if_clear; }
static inline uint32_t mask_select_u32(unsigned long long mask, uint32_t if_set, uint32_t if_clear) { return ((mask >> lane()) & 1ull) ? if_set : if_clear; }
static inline float mask_select0_f32(unsigned long long mask, float if_set) { return ((mask >> lane()) & 1ull) ? if_set : 0.f; }
// top-packed variant (csrc/wave64.h wave_reduce9_swap_f32): lane group g = lane >> 3 receives the total of value bitrev3(g)
static inline int wave_swap9_component(int lane_)
{
	const int g = lane_ >> 3;
	return ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);
}
static inline void wave_reduce9_swap_f32(const float (&v)[9], float& packed, float& ninth_row)
{
	float t[9];
	for (int c = 0; c < 9; c++) t[c] = v[c];
	uint32_t bits;
	memcpy(&bits, &v[8], 4);
	const uint64_t* s = wave_exchange(bits);
	float row = 0.f;
	for (int i = 0; i < 16; i++) {
		uint32_t b = (uint32_t)s[(lane() & ~15) + i];
		float f;
		memcpy(&f, &b, 4);
		row += f;
	}
	wave_sync();
	wave_reduce9_f32(t);
	packed = t[wave_swap9_component(lane())];
	ninth_row = row;
}
// asserts the value really is wave-uniform (the real primitive silently takes lane 0's)
static inline unsigned long long wave_uniform_u64(unsigned long long v)
{
	const uint64_t* s = wave_exchange(v);
	for (int i = 0; i < 64; i++)
		if (s[i] != v) {
			fprintf(stderr, "hipemu: wave_uniform called with a non-uniform value\n");
			abort();
		}
	wave_sync();
	return v;
}
static inline uint32_t wave_uniform_u32(uint32_t v) { return (uint32_t)wave_uniform_u64(v); }
static inline uint32_t wave_readlane_u32(uint32_t v, int src) { return wave_shfl_u32(v, src); }
static inline float wave_readlane_f32(float v, int src)
{
	uint32_t b;
	memcpy(&b, &v, 4);
	b = wave_shfl_u32(b, src);
	float r;
	memcpy(&r, &b, 4);
	return r;
}
// `uniform_value` only needs to be right in lane 63 here (the HIP version reads it from an SGPR
// that was itself filled by readlane(.., 63)); every lane passes its own copy, lane `lane` keeps it.
static inline float wave_writelane_f32(float old, float uniform_value, int dst_lane) { return lane() == dst_lane ? uniform_value : old; }
}  // namespace hipemu

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline void __threadfence() {}   // (blocks run one after another here: every store is visible to the next block)

template <typename T>
static inline T atomicAdd(T* p, T v)
{
	T old = *p;
	*p = old + v;
	return old;
}
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

namespace hipemu {
}  // namespace hipemu
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
