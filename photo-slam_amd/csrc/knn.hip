// knn.hip -- mean squared distance to the 3 nearest neighbours (simple-knn).
//
// Semantics follow SimpleKNN::knn (third_party/simple-knn/simple_knn.cu:185-221): scene AABB
// with {0,0,0} as the initial value of BOTH the min and the max reduction (:191-200), 30-bit
// Morton codes (:45-61), stable sort by code, boxes of 1024 Morton-consecutive points
// (:78-117), then per point an exact 3-NN search that visits only boxes whose AABB distance
// does not exceed the current 3rd-best (:147-183).
//
// Differences in structure, not in results:
//   * no host round trips (the reference blocks on two D2H copies of the AABB and calls
//     cudaMalloc/thrust allocations per invocation): the AABB stays in HBM and scratch
//     comes from the caller;
//   * the sorted points are gathered once into a contiguous array so box scans read consecutive
//     addresses (the reference gathers every candidate through the index array);
//   * a three-level hierarchy of bounding boxes over the Morton order -- 32 points, the reference's
//     1024 points, 32768 points -- prunes the scan: the reference tests every 1024-box against the
//     query (P/1024 tests per point) and scans all 1024 points of a box that can still hold a
//     closer one; here a super box that cannot is skipped with its 32 boxes, and inside a surviving
//     box only the 32-point runs that can are scanned.  The result is the same set of three
//     smallest distances: a box is skipped only if its (float, monotone) lower bound exceeds the
//     current third-best, exactly the reference's test (:165-166) applied at finer grain;
//   * compiled with -ffp-contract=off so distances are bit-identical to the CPU oracle.
#include "state.h"
#include "wave64.h"
#include "kernels.h"

#include <float.h>

namespace gsr {

constexpr int KNN_BOX = 1024;      // BOX_SIZE, simple_knn.cu:10
constexpr int KNN_SUB = 32;        // points per sub-box (Morton-consecutive: a compact cell)
constexpr int KNN_FAN = 32;        // sub-boxes per box, boxes per super box
static_assert(KNN_SUB * KNN_FAN == KNN_BOX, "a box is KNN_FAN sub-boxes");
constexpr int KNN_THREADS = 256;

struct KnnState {
	float* aabb;          // [8]: min xyz, max xyz
	float* partial;       // [nblk][6]
	uint32_t* codes;      // [P]
	uint32_t* keys_a;     // [P]
	uint32_t* vals_a;     // [P]
	uint32_t* keys_b;     // [P]
	uint32_t* vals_b;     // [P]
	float* sorted_pts;    // [3P]
	float* boxes;         // [nbox][6]
	float* subs;          // [nsub][6]  bounding boxes of the runs of KNN_SUB points
	float* supers;        // [nsup][6]  bounding boxes of KNN_FAN consecutive boxes
	uint32_t* sort_scratch;
	static KnnState carve(char* chunk, size_t P, size_t* bytes = nullptr)
	{
		KnnState k;
		Carver c(chunk);
		const size_t nblk = (P + 1023) / 1024 + 1;
		const size_t nbox = (P + KNN_BOX - 1) / KNN_BOX + 1;
		k.aabb = c.take<float>(8);
		k.partial = c.take<float>(6 * nblk);
		k.codes = c.take<uint32_t>(P);
		k.keys_a = c.take<uint32_t>(P);
		k.vals_a = c.take<uint32_t>(P);
		k.keys_b = c.take<uint32_t>(P);
		k.vals_b = c.take<uint32_t>(P);
		k.sorted_pts = c.take<float>(3 * P);
		k.boxes = c.take<float>(6 * nbox);
		k.subs = c.take<float>(6 * ((P + KNN_SUB - 1) / KNN_SUB + 1));
		k.supers = c.take<float>(6 * ((nbox + KNN_FAN - 1) / KNN_FAN + 1));
		k.sort_scratch = c.take<uint32_t>(sort_scratch_elems((int)P));
		if (bytes) *bytes = c.used(chunk) + 128;
		return k;
	}
};

size_t knn_scratch_bytes(int P)
{
	size_t b = 0;
	KnnState::carve(nullptr, (size_t)P, &b);
	return b;
}

// Block-wide min/max of 3 components through LDS; result valid in thread 0.
__device__ __forceinline__ void block_minmax(float (&mn)[3], float (&mx)[3], float (*s)[KNN_THREADS])
{
	const int t = (int)threadIdx.x;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		s[c][t] = mn[c];
		s[3 + c][t] = mx[c];
	}
	__syncthreads();
	for (int off = KNN_THREADS / 2; off >= 1; off >>= 1) {
		if (t < off) {
#pragma unroll
			for (int c = 0; c < 3; c++) {
				s[c][t] = fminf(s[c][t], s[c][t + off]);
				s[3 + c][t] = fmaxf(s[3 + c][t], s[3 + c][t + off]);
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int c = 0; c < 3; c++) {
		mn[c] = s[c][0];
		mx[c] = s[3 + c][0];
	}
}

// cub::DeviceReduce::Reduce with init {0,0,0} for both reductions, simple_knn.cu:191-200.
__global__ void __launch_bounds__(KNN_THREADS)
knn_aabb_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial)
{
	__shared__ float s[6][KNN_THREADS];
	float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
	const int base = (int)blockIdx.x * 1024;
	for (int i = base + (int)threadIdx.x; i < min(P, base + 1024); i += KNN_THREADS) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const float v = pts[3 * (size_t)i + c];
			mn[c] = fminf(mn[c], v);
			mx[c] = fmaxf(mx[c], v);
		}
	}
	block_minmax(mn, mx, s);
	if (threadIdx.x == 0) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			partial[6 * (size_t)blockIdx.x + c] = mn[c];
			partial[6 * (size_t)blockIdx.x + 3 + c] = mx[c];
		}
	}
}
__global__ void __launch_bounds__(KNN_THREADS)
knn_aabb_final_kernel(int nblk, const float* __restrict__ partial, float* __restrict__ aabb)
{
	__shared__ float s[6][KNN_THREADS];
	float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
	for (int i = (int)threadIdx.x; i < nblk; i += KNN_THREADS) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			mn[c] = fminf(mn[c], partial[6 * (size_t)i + c]);
			mx[c] = fmaxf(mx[c], partial[6 * (size_t)i + 3 + c]);
		}
	}
	block_minmax(mn, mx, s);
	if (threadIdx.x == 0) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			aabb[c] = mn[c];
			aabb[4 + c] = mx[c];
		}
	}
}

// prepMorton / coord2Morton, simple_knn.cu:45-70
__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}
// float -> uint32 conversion with device semantics (truncate, saturate, NaN -> 0)
__device__ __forceinline__ uint32_t f2u(float f)
{
	if (!(f > 0.0f)) return 0u;
	if (f >= 4294967296.0f) return 0xFFFFFFFFu;
	return (uint32_t)f;
}
__global__ void __launch_bounds__(KNN_THREADS)
knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ aabb, uint32_t* __restrict__ codes)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= P) return;
	const float mnx = aabb[0], mny = aabb[1], mnz = aabb[2], mxx = aabb[4], mxy = aabb[5], mxz = aabb[6];
	const uint32_t x = prep_morton(f2u(((pts[3 * (size_t)i] - mnx) / (mxx - mnx)) * ((1 << 10) - 1)));
	const uint32_t y = prep_morton(f2u(((pts[3 * (size_t)i + 1] - mny) / (mxy - mny)) * ((1 << 10) - 1)));
	const uint32_t z = prep_morton(f2u(((pts[3 * (size_t)i + 2] - mnz) / (mxz - mnz)) * ((1 << 10) - 1)));
	codes[i] = x | (y << 1) | (z << 2);
}

// Gather points into Morton order and reduce each box of 1024 (boxMinMax, simple_knn.cu:78-117).
__global__ void __launch_bounds__(KNN_THREADS)
knn_gather_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ indices,
                        float* __restrict__ sorted_pts, float* __restrict__ boxes)
{
	__shared__ float s[6][KNN_THREADS];
	float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	const int base = (int)blockIdx.x * KNN_BOX;
	for (int i = base + (int)threadIdx.x; i < min(P, base + KNN_BOX); i += KNN_THREADS) {
		const uint32_t src = indices[i];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const float v = pts[3 * (size_t)src + c];
			sorted_pts[3 * (size_t)i + c] = v;
			mn[c] = fminf(mn[c], v);
			mx[c] = fmaxf(mx[c], v);
		}
	}
	block_minmax(mn, mx, s);
	if (threadIdx.x == 0) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			boxes[6 * (size_t)blockIdx.x + c] = mn[c];
			boxes[6 * (size_t)blockIdx.x + 3 + c] = mx[c];
		}
	}
}

// updateKBest<3>, simple_knn.cu:131-145
__device__ __forceinline__ void update3(float px, float py, float pz, float qx, float qy, float qz, float (&knn)[3])
{
	const float dx = qx - px, dy = qy - py, dz = qz - pz;
	float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
	for (int j = 0; j < 3; j++) {
		if (knn[j] > dist) {
			const float t = knn[j];
			knn[j] = dist;
			dist = t;
		}
	}
}

// Bounding boxes of the runs of KNN_SUB Morton-consecutive points (thread per run: 384 contiguous bytes).
__global__ void __launch_bounds__(KNN_THREADS)
knn_sub_boxes_kernel(int P, const float* __restrict__ spts, float* __restrict__ subs)
{
	const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int first = c * KNN_SUB;
	if (first >= P) return;
	float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	for (int i = first; i < min(P, first + KNN_SUB); i++) {
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const float v = spts[3 * (size_t)i + k];
			mn[k] = fminf(mn[k], v);
			mx[k] = fmaxf(mx[k], v);
		}
	}
#pragma unroll
	for (int k = 0; k < 3; k++) {
		subs[6 * (size_t)c + k] = mn[k];
		subs[6 * (size_t)c + 3 + k] = mx[k];
	}
}
// ... and of KNN_FAN consecutive boxes (thread per super box)
__global__ void __launch_bounds__(KNN_THREADS)
knn_super_boxes_kernel(int nbox, const float* __restrict__ boxes, float* __restrict__ supers)
{
	const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	const int first = s * KNN_FAN;
	if (first >= nbox) return;
	float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
	for (int b = first; b < min(nbox, first + KNN_FAN); b++) {
#pragma unroll
		for (int k = 0; k < 3; k++) {
			mn[k] = fminf(mn[k], boxes[6 * (size_t)b + k]);
			mx[k] = fmaxf(mx[k], boxes[6 * (size_t)b + 3 + k]);
		}
	}
#pragma unroll
	for (int k = 0; k < 3; k++) {
		supers[6 * (size_t)s + k] = mn[k];
		supers[6 * (size_t)s + 3 + k] = mx[k];
	}
}

// distBoxPoint, simple_knn.cu:119-129: a lower bound of the squared distance from p to any point inside the box -- also in
// float arithmetic: every operation is monotone and it is the expression of the point distance (dx dx + dy dy + dz dz, no
// contraction in this translation unit) with |d| replaced by something not larger
__device__ __forceinline__ float box_dist(const float* __restrict__ b, float px, float py, float pz)
{
	float ddx = 0.f, ddy = 0.f, ddz = 0.f;
	if (px < b[0] || px > b[3]) ddx = fminf(fabsf(px - b[0]), fabsf(px - b[3]));
	if (py < b[1] || py > b[4]) ddy = fminf(fabsf(py - b[1]), fabsf(py - b[4]));
	if (pz < b[2] || pz > b[5]) ddz = fminf(fabsf(pz - b[2]), fabsf(pz - b[5]));
	return ddx * ddx + ddy * ddy + ddz * ddz;
}

// boxMeanDist, simple_knn.cu:147-183.  Thread = one point (in Morton order).  Every thread walks the hierarchy in index order
// and descends only where the bound can still improve its 3rd-best (the reference's test `dist > reject || dist > best[2]`,
// :165-166, at three grains); the 64 lanes of a wave are Morton neighbours, so they mostly descend into the same boxes and the
// contiguous, pre-gathered points of a run are fetched once per wave (same-address loads).  No barriers, no LDS.
__global__ void __launch_bounds__(KNN_THREADS)
knn_mean_dist_kernel(int P, const float* __restrict__ spts, const uint32_t* __restrict__ indices,
                     const float* __restrict__ boxes, int nbox, const float* __restrict__ subs, const float* __restrict__ supers,
                     float* __restrict__ dists)
{
	const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (idx >= P) return;
	const float px = spts[3 * (size_t)idx], py = spts[3 * (size_t)idx + 1], pz = spts[3 * (size_t)idx + 2];
	float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
	for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
		if (i == idx) continue;
		update3(px, py, pz, spts[3 * (size_t)i], spts[3 * (size_t)i + 1], spts[3 * (size_t)i + 2], best);
	}
	const float reject = best[2];
	best[0] = FLT_MAX;
	best[1] = FLT_MAX;
	best[2] = FLT_MAX;
	const int nsup = (nbox + KNN_FAN - 1) / KNN_FAN, nsub = (P + KNN_SUB - 1) / KNN_SUB;
	for (int s = 0; s < nsup; s++) {
		const float ds = box_dist(supers + 6 * (size_t)s, px, py, pz);
		if (ds > reject || ds > best[2]) continue;
		const int b_end = min(nbox, (s + 1) * KNN_FAN);
		for (int b = s * KNN_FAN; b < b_end; b++) {
			const float db = box_dist(boxes + 6 * (size_t)b, px, py, pz);
			if (db > reject || db > best[2]) continue;
			const int c_end = min(nsub, (b + 1) * KNN_FAN);
			for (int c = b * KNN_FAN; c < c_end; c++) {
				const float dc = box_dist(subs + 6 * (size_t)c, px, py, pz);
				if (dc > reject || dc > best[2]) continue;
				const int i_end = min(P, (c + 1) * KNN_SUB);
				for (int i = c * KNN_SUB; i < i_end; i++) {
					if (i == idx) continue;
					update3(px, py, pz, spts[3 * (size_t)i], spts[3 * (size_t)i + 1], spts[3 * (size_t)i + 2], best);
				}
			}
		}
	}
	dists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

float* morton_points_buffer(int n, char* scratch) { return KnnState::carve(scratch, (size_t)n).sorted_pts; }

int launch_morton_order(int n, const float* points, uint32_t** order_out, char* scratch, hipStream_t stream)
{
	KnnState k = KnnState::carve(scratch, (size_t)n);
	const int nblk = div_up(n, 1024);
	GSR_LAUNCH(knn_aabb_partial_kernel, nblk, KNN_THREADS, stream, n, points, k.partial);
	GSR_LAUNCH(knn_aabb_final_kernel, 1, KNN_THREADS, stream, nblk, (const float*)k.partial, k.aabb);
	GSR_LAUNCH(knn_morton_kernel, div_up(n, KNN_THREADS), KNN_THREADS, stream, n, points, (const float*)k.aabb, k.codes);
	uint32_t* kres = nullptr;
	const int st = launch_radix_sort(k.codes, nullptr, k.keys_a, k.vals_a, k.keys_b, k.vals_b, n, 0, 32, k.sort_scratch, stream, &kres, order_out);
	if (st != GSR_OK) return st;
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_knn(int P, const float* points, float* meanDists, char* scratch, hipStream_t stream)
{
	KnnState k = KnnState::carve(scratch, (size_t)P);
	const int nblk = div_up(P, 1024);
	const int nbox = div_up(P, KNN_BOX);
	GSR_LAUNCH(knn_aabb_partial_kernel, nblk, KNN_THREADS, stream, P, points, k.partial);
	GSR_LAUNCH(knn_aabb_final_kernel, 1, KNN_THREADS, stream, nblk, (const float*)k.partial, k.aabb);
	GSR_LAUNCH(knn_morton_kernel, div_up(P, KNN_THREADS), KNN_THREADS, stream, P, points, (const float*)k.aabb, k.codes);
	uint32_t *kres = nullptr, *vres = nullptr;
	int st = launch_radix_sort(k.codes, nullptr, k.keys_a, k.vals_a, k.keys_b, k.vals_b, P, 0, 32, k.sort_scratch, stream,
	                           &kres, &vres);
	if (st != GSR_OK) return st;
	GSR_LAUNCH(knn_gather_boxes_kernel, nbox, KNN_THREADS, stream, P, points, (const uint32_t*)vres, k.sorted_pts, k.boxes);
	GSR_LAUNCH(knn_sub_boxes_kernel, div_up(div_up(P, KNN_SUB), KNN_THREADS), KNN_THREADS, stream, P, (const float*)k.sorted_pts, k.subs);
	GSR_LAUNCH(knn_super_boxes_kernel, div_up(div_up(nbox, KNN_FAN), KNN_THREADS), KNN_THREADS, stream, nbox, (const float*)k.boxes, k.supers);
	GSR_LAUNCH(knn_mean_dist_kernel, div_up(P, KNN_THREADS), KNN_THREADS, stream, P, (const float*)k.sorted_pts,
	           (const uint32_t*)vres, (const float*)k.boxes, nbox, (const float*)k.subs, (const float*)k.supers, meanDists);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
