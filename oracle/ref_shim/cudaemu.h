/*
 * cudaemu.h -- just enough of the CUDA execution model to compile the REFERENCE's own rasterizer sources
 * (cuda_rasterizer/{forward,backward,rasterizer_impl}.cu) for the host with g++.  TEST INFRASTRUCTURE ONLY
 * (see oracle/build_ref.py): the product never links it.
 *
 * Model: a kernel launch runs its thread blocks one after the other; the threads of a block are ucontext fibers
 * that run one at a time and switch only at block barriers (__syncthreads, __syncthreads_count,
 * cooperative_groups::thread_block::sync), so atomics are trivially atomic and execution is deterministic (the
 * reference's float atomicAdd order becomes: block order, then thread order).  __shared__ is a function-local static.
 */
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

namespace cudaemu {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void syncthreads();
int syncthreads_count(int pred);
}  // namespace cudaemu
static inline void __syncthreads() { cudaemu::syncthreads(); }
static inline int __syncthreads_count(int pred) { return cudaemu::syncthreads_count(pred); }
static inline void __trap() { abort(); }

/* CUDA's mixed-signedness min/max overloads (device functions) */
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
static inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
static inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
static inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }

/* runtime API subset (host memory is "device" memory) */
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memmove(d, s, n); return cudaSuccess; }
template <typename T>
static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)malloc(n); return *p ? cudaSuccess : 2; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* p, int v, size_t n) { memset(p, v, n); return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return "cudaemu"; }

/* kernel<<<grid, block>>>(args...) is rewritten by oracle/build_ref.py into CUDAEMU_LAUNCH((kernel), grid, block, args...) */
#define CUDAEMU_LAUNCH(kernel, grid, block, ...) ::cudaemu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
