/* shim of the part of <cooperative_groups.h> the reference rasterizer uses (test infrastructure, see ../cudaemu.h) */
#pragma once
#include "../cudaemu.h"

namespace cooperative_groups {
struct grid_group {
	/* the reference only launches 1-D grids of 1-D blocks through this_grid() */
	unsigned long long thread_rank() const
	{
		const unsigned long long block = ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
		const unsigned long long in_block = ((unsigned long long)threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x;
		return block * ((unsigned long long)blockDim.x * blockDim.y * blockDim.z) + in_block;
	}
};
struct thread_block {
	dim3 group_index() const { return dim3(blockIdx.x, blockIdx.y, blockIdx.z); }
	dim3 thread_index() const { return dim3(threadIdx.x, threadIdx.y, threadIdx.z); }
	unsigned thread_rank() const { return (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x; }
	void sync() const { cudaemu::syncthreads(); }
};
static inline grid_group this_grid() { return grid_group(); }
static inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
