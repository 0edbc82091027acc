/* shim of the CUB device algorithms the reference rasterizer and simple-knn call (NVIDIA/cub, header-only, shipped with the CUDA
 * toolkit the reference builds against; not vendored in the reference tree).  Published semantics restated on the host:
 *   DeviceScan::InclusiveSum      out[i] = in[0] + ... + in[i]
 *   DeviceRadixSort::SortPairs    STABLE ascending sort of (key, value) pairs on key bits [begin_bit, end_bit)
 *   DeviceReduce::Reduce          out[0] = op(...op(op(init, in[0]), in[1])..., in[n-1])  (the callers' ops, min / max
 *                                 per component, do not depend on the order)
 * Both follow CUB's two-call protocol: d_temp_storage == nullptr -> only report temp_storage_bytes. */
#pragma once
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <type_traits>
#include <vector>

#include "../../cudaemu.h"

namespace cub {
struct DeviceScan {
	template <typename InT, typename OutT>
	static cudaError_t InclusiveSum(void* d_temp_storage, size_t& temp_storage_bytes, InT d_in, OutT d_out, int num_items)
	{
		if (d_temp_storage == nullptr) {
			temp_storage_bytes = 1024;
			return cudaSuccess;
		}
		/* element type of the output, wrap-around arithmetic like the device code */
		typename std::remove_reference<decltype(d_out[0])>::type acc = 0;
		for (int i = 0; i < num_items; i++) {
			acc += d_in[i];
			d_out[i] = acc;
		}
		return cudaSuccess;
	}
};
struct DeviceReduce {
	template <typename InT, typename OutT, typename Op, typename T>
	static cudaError_t Reduce(void* d_temp_storage, size_t& temp_storage_bytes, InT d_in, OutT d_out, int num_items, Op op, T init)
	{
		if (d_temp_storage == nullptr) {
			temp_storage_bytes = 1024;
			return cudaSuccess;
		}
		T acc = init;
		for (int i = 0; i < num_items; i++) acc = op(acc, d_in[i]);
		d_out[0] = acc;
		return cudaSuccess;
	}
};
struct DeviceRadixSort {
	template <typename KeyT, typename ValueT>
	static cudaError_t SortPairs(void* d_temp_storage, size_t& temp_storage_bytes, const KeyT* d_keys_in, KeyT* d_keys_out,
	                             const ValueT* d_values_in, ValueT* d_values_out, int num_items, int begin_bit = 0,
	                             int end_bit = sizeof(KeyT) * 8)
	{
		if (d_temp_storage == nullptr) {
			temp_storage_bytes = 1024;
			return cudaSuccess;
		}
		const int nbits = end_bit - begin_bit;
		const KeyT mask = nbits >= (int)(sizeof(KeyT) * 8) ? ~KeyT(0) : ((KeyT(1) << nbits) - 1);
		std::vector<int> idx(num_items);
		std::iota(idx.begin(), idx.end(), 0);
		std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
			return ((d_keys_in[a] >> begin_bit) & mask) < ((d_keys_in[b] >> begin_bit) & mask);
		});
		for (int i = 0; i < num_items; i++) {
			d_keys_out[i] = d_keys_in[idx[i]];
			d_values_out[i] = d_values_in[idx[i]];
		}
		return cudaSuccess;
	}
};
}  // namespace cub
