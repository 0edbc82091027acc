// rocPRIM's device radix sort (onesweep on this architecture) on the three sorts of a C3 forward pass -- a reference point for
// csrc/sort.hip: 32-bit depth keys of the V visible Gaussians, the same over all P, and the 13 tile bits of
// the R instances.  hipcc --offload-arch=gfx950 -O3 tools/rocprim_sort_probe.hip -o /tmp/rocprim_sort_probe
#include <cstring>   // (rocPRIM's texture iterator header calls memset without including it)
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static int run(const char* name, size_t n, unsigned begin_bit, unsigned end_bit, bool float_like)
{
	std::vector<unsigned> hk(n), hv(n);
	unsigned s = 12345u;
	for (size_t i = 0; i < n; i++) {
		s = s * 1664525u + 1013904223u;
		if (float_like) {   // positive depths 0.2 .. 20 as float bits
			float d = 0.2f + 19.8f * (float)(s >> 8) / 16777216.0f;
			unsigned b; memcpy(&b, &d, 4); hk[i] = b;
		} else hk[i] = (s >> 8) % 8160u;
		hv[i] = (unsigned)i;
	}
	unsigned *ki, *ko, *vi, *vo;
	CK(hipMalloc(&ki, n * 4)); CK(hipMalloc(&ko, n * 4)); CK(hipMalloc(&vi, n * 4)); CK(hipMalloc(&vo, n * 4));
	CK(hipMemcpy(ki, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vi, hv.data(), n * 4, hipMemcpyHostToDevice));
	size_t tmp_bytes = 0;
	CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, begin_bit, end_bit, 0));
	void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int w = 0; w < 5; w++) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, begin_bit, end_bit, 0));
	CK(hipDeviceSynchronize());
	const int reps = 50;
	CK(hipEventRecord(a, 0));
	for (int r = 0; r < reps; r++) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, begin_bit, end_bit, 0));
	CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
	float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
	// stability + order check against std::stable_sort
	std::vector<unsigned> ok(n), ov(n);
	CK(hipMemcpy(ok.data(), ko, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), vo, n * 4, hipMemcpyDeviceToHost));
	std::vector<unsigned> idx(n);
	for (size_t i = 0; i < n; i++) idx[i] = (unsigned)i;
	const unsigned mask = end_bit - begin_bit >= 32 ? 0xFFFFFFFFu : (((1u << (end_bit - begin_bit)) - 1u) << begin_bit);
	std::stable_sort(idx.begin(), idx.end(), [&](unsigned x, unsigned y) { return (hk[x] & mask) < (hk[y] & mask); });
	size_t bad = 0;
	for (size_t i = 0; i < n; i++) bad += ov[i] != idx[i];
	printf("{\"sort\": \"%s\", \"n\": %zu, \"bits\": \"%u..%u\", \"us_per_sort\": %.1f, \"temp_MB\": %.1f, \"stable_and_sorted\": %s}\n", name, n,
	       begin_bit, end_bit, 1e3f * ms / reps, tmp_bytes / 1e6, bad == 0 ? "true" : "false");
	(void)hipFree(ki); (void)hipFree(ko); (void)hipFree(vi); (void)hipFree(vo); (void)hipFree(tmp);
	return 0;
}

int main()
{
	if (run("depth keys of the visible Gaussians (V)", 934432, 0, 32, true)) return 1;
	if (run("depth keys of all Gaussians (P)", 2000000, 0, 32, true)) return 1;
	if (run("tile bits of the instances (R)", 6237787, 0, 13, false)) return 1;
	return 0;
}
