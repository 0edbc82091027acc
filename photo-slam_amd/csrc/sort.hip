// sort.hip -- prefix scan and stable LSD radix sort for gfx950 (wave64).
//
// Replaces cub::DeviceScan::InclusiveSum (cuda_rasterizer/rasterizer_impl.cu:276) and
// cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:303-308, simple_knn.cu:210-213).
// Both are HBM-streaming integer kernels: no MFMA, loads/stores coalesced per wave,
// ranking by wave ballots (64-bit masks) instead of shared-memory atomics, and the
// scatter goes through an LDS-staged locally sorted tile so that every digit run leaves
// the workgroup as one contiguous burst.
#include "state.h"
#include "wave64.h"

namespace gsr {

// ------------------------------------------------------------------ scan
// (block_excl_scan_256: wave64.h)
__global__ void __launch_bounds__(SCAN_THREADS)
scan_reduce_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather, uint32_t* __restrict__ staged,
                   uint32_t* __restrict__ block_sums, int n, int items_per_block, const uint32_t* __restrict__ n_dev)
{
	// n_dev (nullable): only the first *n_dev elements exist (the rest of the gather list is undefined): they count as zeros
	__shared__ uint32_t s_wave[4];
	const int base = blockIdx.x * items_per_block;
	const int end_all = min(n, base + items_per_block);
	const int end = n_dev ? min(end_all, (int)*n_dev) : end_all;
	uint32_t acc = 0;
	if (staged && n_dev)
		for (int i = max(base, end) + (int)threadIdx.x; i < end_all; i += SCAN_THREADS) staged[i] = 0u;
	// a gathered input (in[gather[i]]: one 64-byte line per element, measured 10x the linear traffic) is staged in the output
	// array here, so that the apply pass reads it linearly instead of gathering a second time
	for (int i = base + (int)threadIdx.x; i < end; i += SCAN_THREADS) {
		const uint32_t v = gather ? in[gather[i]] : in[i];
		if (staged) staged[i] = v;
		acc += v;
	}
	uint32_t tot;
	block_excl_scan_256(acc, &tot, s_wave);
	if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// The same first pass for the offsets of the instance emission: the element is the tile count of the rectangle of Gaussian
// gather[i] -- ONE random gather serves the scan (the count) and the emission (the rectangle, left in rect_sorted in scan
// order) where there used to be two, of tiles_touched here and of the rectangle there (60 MB of 64-byte lines each at C3).
__global__ void __launch_bounds__(SCAN_THREADS)
scan_reduce_rect_kernel(const uint2* __restrict__ rect, const uint32_t* __restrict__ gather, uint32_t* __restrict__ staged,
                        uint2* __restrict__ rect_sorted, uint32_t* __restrict__ block_sums, int n, int items_per_block,
                        const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ long_runs, uint32_t* __restrict__ long_counts,
                        uint32_t long_capacity)
{
	__shared__ uint32_t s_wave[4];
	const int base = blockIdx.x * items_per_block;
	const int end_all = min(n, base + items_per_block);
	const int end = n_dev ? min(end_all, (int)*n_dev) : end_all;
	uint32_t acc = 0;
	if (n_dev)
		for (int i = max(base, end) + (int)threadIdx.x; i < end_all; i += SCAN_THREADS) staged[i] = 0u;
	for (int i0 = base; i0 < end; i0 += SCAN_THREADS) {   // (block-uniform trip count: the ballot below is a wave's)
		const int i = i0 + (int)threadIdx.x;
		uint32_t id = 0u, v = 0u;
		if (i < end) {
			id = gather[i];
			const uint2 r = rect[id];
			v = ((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16));
			rect_sorted[i] = r;
			staged[i] = v;
			acc += v;
		}
		// Gaussians whose run of instance slots is too long for one lane of the backward preprocess (state.h: LONG_RUN): listed
		// here, one atomic per wave that has any (a few thousand entries at C3); the list order is irrelevant to the results
		const unsigned long long lm = long_runs ? wave_ballot(v > LONG_RUN) : 0ull;
		if (lm) {
			const int leader = __ffsll((long long)lm) - 1;
			uint32_t at = 0;
			// (the scan runs in DEPTH order and the long runs are the near Gaussians: they sit in its first few blocks -- a sub-list
			// per block left long_run_sums_kernel with all of them in a handful of sub-lists, 110 us instead of 20.  Every (block,
			// trip, wave) takes the next sub-list instead.)
			const uint32_t list = (((uint32_t)blockIdx.x * (uint32_t)(items_per_block / SCAN_THREADS) + (uint32_t)((i0 - base) / SCAN_THREADS)) * (SCAN_THREADS / 64) +
			                       (uint32_t)wave_id()) % (uint32_t)LONG_LISTS;
			if (lane_id() == leader) at = atomicAdd(&long_counts[list * LONG_COUNT_STRIDE], (uint32_t)__popcll(lm));
			at = wave_shfl_u32(at, leader);
			if ((lm >> lane_id()) & 1ull) long_runs[(size_t)list * long_capacity + at + (uint32_t)__popcll(lm & lanemask_lt())] = id;
		}
	}
	uint32_t tot;
	block_excl_scan_256(acc, &tot, s_wave);
	if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// seeds (nullable; exclusive scans of counts only): seeds[k] = the element whose run [out[i], out[i] + v) holds position
// k * seed_stride -- the instance emission starts its walk there instead of searching the offsets (binning.hip).
__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(const uint32_t* in, const uint32_t* __restrict__ gather, uint32_t* out,   // in may alias out (staged input)
                  const uint32_t* __restrict__ block_sums, int n, int items_per_block, int inclusive,
                  uint32_t* __restrict__ seeds, uint32_t seed_stride, uint32_t seed_capacity)
{
	__shared__ uint32_t s_wave[4];
	const int base = blockIdx.x * items_per_block;
	const int end = min(n, base + items_per_block);
	// the block's carry = the sum of the block sums in front of it, formed HERE (a few hundred words, one block reduction)
	// instead of by a single-workgroup launch between the two passes: one launch and its bubble less per scan
	uint32_t carry;
	{
		uint32_t part = 0;
		for (int j = (int)threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) part += block_sums[j];
		(void)block_excl_scan_256(part, &carry, s_wave);
	}
	if (!gather) {
		// eight consecutive elements per thread and trip (two 16-byte loads, one block scan per 2 048 elements -- it was a block scan
		// with its two barriers per 256: 8.8 -> about 5 us for the 2 M offsets of C3)
		constexpr int IPT = 8;
		const bool aligned16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
		for (int b = base; b < end; b += SCAN_THREADS * IPT) {
			const int i0 = b + (int)threadIdx.x * IPT;
			uint32_t v[IPT];
			const bool vec = i0 + IPT <= end && aligned16;   // (b is a multiple of 2 048 elements: the vectors are as aligned as the arrays)
			if (vec) {
				const uint4 lo = *reinterpret_cast<const uint4*>(in + i0), hi = *reinterpret_cast<const uint4*>(in + i0 + 4);
				v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
			} else {
#pragma unroll
				for (int j = 0; j < IPT; j++) v[j] = i0 + j < end ? in[i0 + j] : 0u;
			}
			uint32_t sum = 0;
#pragma unroll
			for (int j = 0; j < IPT; j++) sum += v[j];
			uint32_t tot;
			uint32_t run = carry + block_excl_scan_256(sum, &tot, s_wave);   // exclusive prefix of this thread's first element
			uint32_t o[IPT];
#pragma unroll
			for (int j = 0; j < IPT; j++) {
				const uint32_t first = run;
				run += v[j];
				o[j] = inclusive ? run : first;
				if (seeds && i0 + j < end && v[j] != 0u)
					// (one element in seed_stride / mean count holds a seed position; a screen-filling splat holds a few dozen)
					for (uint32_t k = (first + seed_stride - 1u) / seed_stride; k < seed_capacity && k * seed_stride < first + v[j]; k++) seeds[k] = (uint32_t)(i0 + j);
			}
			if (vec) {
				*reinterpret_cast<uint4*>(out + i0) = make_uint4(o[0], o[1], o[2], o[3]);
				*reinterpret_cast<uint4*>(out + i0 + 4) = make_uint4(o[4], o[5], o[6], o[7]);
			} else {
#pragma unroll
				for (int j = 0; j < IPT; j++)
					if (i0 + j < end) out[i0 + j] = o[j];
			}
			carry += tot;
		}
		return;
	}
	for (int b = base; b < end; b += SCAN_THREADS) {
		const int i = b + (int)threadIdx.x;
		const uint32_t v = i < end ? in[gather[i]] : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan_256(v, &tot, s_wave);
		if (i < end) out[i] = carry + ex + (inclusive ? v : 0u);
		if (seeds && i < end && v != 0u) {
			const uint32_t first = carry + ex;
			for (uint32_t k = (first + seed_stride - 1u) / seed_stride; k < seed_capacity && k * seed_stride < first + v; k++) seeds[k] = (uint32_t)i;
		}
		carry += tot;
	}
}

int launch_scan_rect_tiles(const uint2* rect, const uint32_t* gather, uint32_t* out, uint2* rect_sorted, int n, uint32_t* scratch,
                           hipStream_t stream, const uint32_t* n_dev, uint32_t* seeds, uint32_t seed_stride, uint32_t seed_capacity,
                           uint32_t* long_runs, uint32_t* long_counts, uint32_t long_capacity)
{
	if (!rect || !gather || !out || !rect_sorted) return GSR_ERR_INVALID_ARG;
	if (n <= 0) return GSR_OK;
	const int ipb = scan_items_per_block(n);
	const int nb = div_up(n, ipb);
	GSR_LAUNCH(scan_reduce_rect_kernel, nb, SCAN_THREADS, stream, rect, gather, out, rect_sorted, scratch, n, ipb, n_dev, long_runs, long_counts,
	           long_capacity);
	GSR_LAUNCH(scan_apply_kernel, nb, SCAN_THREADS, stream, (const uint32_t*)out, (const uint32_t*)nullptr, out,
	           (const uint32_t*)scratch, n, ipb, 0, seeds, seed_stride, seed_capacity);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int launch_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int n, bool inclusive,
                    uint32_t* scratch, hipStream_t stream, const uint32_t* n_dev)
{
	// n_dev (nullable, needs `gather`): elements from *n_dev on count as zeros (their gather indices are undefined)
	if (n_dev && !gather) return GSR_ERR_INVALID_ARG;
	if (n <= 0) return GSR_OK;
	const int ipb = scan_items_per_block(n);
	const int nb = div_up(n, ipb);
	uint32_t* staged = gather ? out : nullptr;
	GSR_LAUNCH(scan_reduce_kernel, nb, SCAN_THREADS, stream, in, gather, staged, scratch, n, ipb, n_dev);
	GSR_LAUNCH(scan_apply_kernel, nb, SCAN_THREADS, stream, gather ? (const uint32_t*)out : in, (const uint32_t*)nullptr, out,
	           (const uint32_t*)scratch, n, ipb, inclusive ? 1 : 0, (uint32_t*)nullptr, 1u, 0u);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// the 64-bit wave sum from 32-bit wave sums of three pieces of every lane's value (64 lanes x 2^16 fits 32 bits; the forward
// pass's instance total decides "more than 2^31 instances": it must not wrap)
__device__ __forceinline__ unsigned long long wave_sum_u64_by_parts(unsigned long long t)
{
	const uint32_t lo = (uint32_t)(t & 0xFFFFFFFFull), hi = (uint32_t)(t >> 32);
	return (unsigned long long)wave_sum_u32(lo & 0xFFFFu) + ((unsigned long long)wave_sum_u32(lo >> 16) << 16) +
	       ((unsigned long long)wave_sum_u32(hi) << 32);
}

// ------------------------------------------------------------------ tile-first binning: the visible Gaussians compacted in id order
// (gsr_api.hip: binning_tile_first).  The depth-first arrangement sorts the Gaussians by depth before their instances are emitted --
// nine launches whatever the size, 56 us for the 17 k visible Gaussians of a 50 k-Gaussian map.  Here the visible Gaussians are only
// COMPACTED (ascending id) and scanned; the instances then reach their tiles in id order (the tile sort is stable) and every tile's
// list is sorted by depth on its own (tile_depth_sort.hip) -- same final order: (tile, depth bits, id).

// the block sum of the projection kernel's per-wave counts: (tiles touched as 64 bits, visible, largest depth key)
__device__ __forceinline__ void block_sum_counts(unsigned long long t, uint32_t v, uint32_t mx, unsigned long long& T, uint32_t& V, uint32_t& M)
{
	__shared__ unsigned long long s_t[SCAN_THREADS / 64];
	__shared__ uint32_t s_v[SCAN_THREADS / 64], s_m[SCAN_THREADS / 64];
	const unsigned long long wt = wave_sum_u64_by_parts(t);
	const uint32_t wv = wave_sum_u32(v), wm = wave_max_u32(mx);
	__syncthreads();   // (reuse across calls)
	if (lane_id() == 0) {
		s_t[wave_id()] = wt;
		s_v[wave_id()] = wv;
		s_m[wave_id()] = wm;
	}
	__syncthreads();
	T = 0ull; V = 0u; M = 0u;
	for (int i = 0; i < SCAN_THREADS / 64; i++) {
		T += s_t[i];
		V += s_v[i];
		M = max(M, s_m[i]);
	}
}
// entry i of a count table: level 0 = the projection kernel's per-wave pairs (tiles, visible, largest key, -), level 1 = block
// sums of those (tiles lo, tiles hi, visible, largest key)
__device__ __forceinline__ void add_count_entry(const uint4 c, int level, unsigned long long& t, uint32_t& v, uint32_t& mx)
{
	if (level == 0) {
		t += c.x; v += c.y; mx = max(mx, c.z);
	} else {
		t += (unsigned long long)c.x | ((unsigned long long)c.y << 32); v += c.z; mx = max(mx, c.w);
	}
}

// block b: the sum of ITS `pairs_per_block` per-wave pairs -> partials[b] (models of more than 128 k Gaussians: two levels)
__global__ void __launch_bounds__(SCAN_THREADS)
count_partials_kernel(const uint4* __restrict__ pairs, int n_pairs, int pairs_per_block, uint4* __restrict__ partials)
{
	const int lo = (int)blockIdx.x * pairs_per_block, hi = min(n_pairs, lo + pairs_per_block);
	unsigned long long t = 0ull;
	uint32_t v = 0u, mx = 0u;
	for (int i = lo + (int)threadIdx.x; i < hi; i += SCAN_THREADS) add_count_entry(pairs[i], 0, t, v, mx);
	unsigned long long T;
	uint32_t V, M;
	block_sum_counts(t, v, mx, T, V, M);
	if (threadIdx.x == 0) partials[blockIdx.x] = make_uint4((uint32_t)(T & 0xFFFFFFFFull), (uint32_t)(T >> 32), V, M);
}

// ONE workgroup: the totals of a count table -> mapped host memory (state.h: HOST_COUNT_WORDS); the event the host waits for is
// recorded behind this launch, and the compaction that follows needs nothing from the host: it covers the host's reaction time
__global__ void __launch_bounds__(SCAN_THREADS)
host_count_kernel(const uint4* __restrict__ entries, int n, int level, uint32_t* __restrict__ host_out)
{
	unsigned long long t = 0ull;
	uint32_t v = 0u, mx = 0u;
	for (int i = (int)threadIdx.x; i < n; i += SCAN_THREADS) add_count_entry(entries[i], level, t, v, mx);
	unsigned long long T;
	uint32_t V, M;
	block_sum_counts(t, v, mx, T, V, M);
	if (threadIdx.x == 0) {
		host_out[0] = (uint32_t)(T & 0xFFFFFFFFull);
		host_out[1] = (uint32_t)(T >> 32);
		host_out[2] = V;
		host_out[3] = M;
	}
}

// Compaction + offsets in ONE pass over the ids: block b owns ids [b ipb, (b + 1) ipb); its carry -- visible Gaussians and tiles
// in front of it -- is the sum of the count-table entries in front of it (`sums`: `epb` entries per block at `level`), so no
// pass over tiles_touched precedes this one.  For the r-th visible Gaussian (ascending id): order[r] = id, offsets[r] = the
// exclusive sum of the tile counts, rect_sorted[r] = its rectangle; seeds / long runs as scan_apply_kernel / scan_reduce_rect_kernel.
__global__ void __launch_bounds__(SCAN_THREADS)
compact_scan_kernel(const uint32_t* __restrict__ tiles_touched, const uint2* __restrict__ rect, const uint4* __restrict__ sums, int level, int epb,
                    uint32_t* __restrict__ order, uint32_t* __restrict__ offsets, uint2* __restrict__ rect_sorted, int n, int items_per_block,
                    uint32_t* __restrict__ seeds, uint32_t seed_stride, uint32_t seed_capacity, uint32_t* __restrict__ long_runs,
                    uint32_t* __restrict__ long_counts, uint32_t long_capacity, uint32_t* __restrict__ visible_out)
{
	__shared__ uint32_t s_wave[4];
	constexpr int IPT = 8, TRIP = SCAN_THREADS * IPT;
	const int base = (int)blockIdx.x * items_per_block;
	const int end = min(n, base + items_per_block);
	uint32_t carry_t, carry_v;
	{
		unsigned long long t = 0ull;
		uint32_t v = 0u, mx = 0u;
		for (int i = (int)threadIdx.x; i < (int)blockIdx.x * epb; i += SCAN_THREADS) add_count_entry(sums[i], level, t, v, mx);
		unsigned long long T;
		uint32_t M;
		block_sum_counts(t, v, mx, T, carry_v, M);
		carry_t = (uint32_t)T;   // (offsets are 32-bit: a view of more than 2^31 instances is refused by the host, which sees the 64-bit total)
	}
	const bool aligned16 = (reinterpret_cast<uintptr_t>(tiles_touched) & 15) == 0;
	for (int b = base; b < end; b += TRIP) {   // (block-uniform trip count: the ballots below are a wave's)
		const int i0 = b + (int)threadIdx.x * IPT;
		uint32_t v[IPT];
		if (i0 + IPT <= end && aligned16) {
			const uint4 lo = *reinterpret_cast<const uint4*>(tiles_touched + i0), hi = *reinterpret_cast<const uint4*>(tiles_touched + i0 + 4);
			v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
		} else {
#pragma unroll
			for (int j = 0; j < IPT; j++) v[j] = i0 + j < end ? tiles_touched[i0 + j] : 0u;
		}
		uint32_t sum = 0, cnt = 0;
#pragma unroll
		for (int j = 0; j < IPT; j++) {
			sum += v[j];
			cnt += v[j] != 0u ? 1u : 0u;
		}
		uint32_t tot_t, tot_v;
		uint32_t run = carry_t + block_excl_scan_256(sum, &tot_t, s_wave);
		uint32_t r = carry_v + block_excl_scan_256(cnt, &tot_v, s_wave);
#pragma unroll
		for (int j = 0; j < IPT; j++) {
			const uint32_t id = (uint32_t)(i0 + j);
			// Gaussians whose run of instance slots is too long for one lane of the backward preprocess (state.h: LONG_RUN): one atomic
			// per wave that has any; every (block, trip, wave, j) appends to the next sub-list (as scan_reduce_rect_kernel)
			const unsigned long long lm = long_runs ? wave_ballot(v[j] > LONG_RUN) : 0ull;
			if (lm) {
				const int leader = __ffsll((long long)lm) - 1;
				uint32_t at = 0;
				const uint32_t list = ((((uint32_t)b / (uint32_t)TRIP) * (SCAN_THREADS / 64) + (uint32_t)wave_id()) * (uint32_t)IPT + (uint32_t)j) % (uint32_t)LONG_LISTS;
				if (lane_id() == leader) at = atomicAdd(&long_counts[list * LONG_COUNT_STRIDE], (uint32_t)__popcll(lm));
				at = wave_shfl_u32(at, leader);
				if ((lm >> lane_id()) & 1ull) long_runs[(size_t)list * long_capacity + at + (uint32_t)__popcll(lm & lanemask_lt())] = id;
			}
			if (v[j] != 0u) {
				order[r] = id;
				offsets[r] = run;
				rect_sorted[r] = rect[id];
				if (seeds)
					for (uint32_t k = (run + seed_stride - 1u) / seed_stride; k < seed_capacity && k * seed_stride < run + v[j]; k++) seeds[k] = r;
				r++;
				run += v[j];
			}
		}
		carry_t += tot_t;
		carry_v += tot_v;
	}
	if (visible_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *visible_out = carry_v;
}

int launch_compact_visible(const uint32_t* tiles_touched, const uint2* rect, const uint4* wave_counts, int n_pairs, uint4* partials,
                           uint32_t* order, uint32_t* offsets, uint2* rect_sorted, int n, uint32_t* host_out, void* ready, hipStream_t stream,
                           uint32_t* seeds, uint32_t seed_stride, uint32_t seed_capacity, uint32_t* long_runs, uint32_t* long_counts,
                           uint32_t long_capacity, uint32_t* visible_out)
{
	if (!tiles_touched || !rect || !wave_counts || !partials || !order || !offsets || !rect_sorted || !host_out) return GSR_ERR_INVALID_ARG;
	if (n <= 0) return GSR_OK;
	const int ipb = scan_items_per_block(n);   // (a multiple of 2 048 = 32 waves of the projection kernel; at most 1 024 blocks)
	const int nb = div_up(n, ipb);
	const int ppb = ipb / 64;
	const bool two_levels = n_pairs > COMPACT_ONE_LEVEL_PAIRS;
	if (two_levels) {
		GSR_LAUNCH(count_partials_kernel, nb, SCAN_THREADS, stream, wave_counts, n_pairs, ppb, partials);
		GSR_LAUNCH(host_count_kernel, 1, SCAN_THREADS, stream, (const uint4*)partials, nb, 1, host_out);
	} else
		GSR_LAUNCH(host_count_kernel, 1, SCAN_THREADS, stream, wave_counts, n_pairs, 0, host_out);
	if (ready) GSR_HIP(hipEventRecord((hipEvent_t)ready, stream));
	GSR_LAUNCH(compact_scan_kernel, nb, SCAN_THREADS, stream, tiles_touched, rect, two_levels ? (const uint4*)partials : wave_counts,
	           two_levels ? 1 : 0, two_levels ? 1 : ppb, order, offsets, rect_sorted, n, ipb, seeds, seed_stride, seed_capacity, long_runs,
	           long_counts, long_capacity, visible_out);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// ------------------------------------------------------------------ radix sort
// Pass structure (per 8-bit digit):  histogram -> exclusive scan of the [bin][block] table
// -> scatter.  Block b always owns elements [b*SORT_CHUNK, (b+1)*SORT_CHUNK) (2 048); wave w of the block owns
// the 512-element sub-range starting at w*512 and walks it in 8 rounds of 64 lane-
// consecutive elements, so the original order inside a digit is (wave, round, lane).

// BINS: 256 (digits of up to 8 bits) or 512 (9 bits: the depth sort's three passes over bias-subtracted keys).  bias is subtracted
// from every key in front of the digit (0 for everybody else): an order-preserving shift that makes the high bits zero.
template <int BINS>
__global__ void __launch_bounds__(SORT_THREADS)
radix_hist_kernel(const uint32_t* __restrict__ keys, int n, int shift, int nbits, uint32_t* __restrict__ hist, int nblocks,
                  const uint32_t* __restrict__ n_dev, int skip_invalid, const RadixHostCount hc, uint32_t bias)
{
	// The forward pass's count for the host rides in the first two launches of the depth sort (gsr_api.hip): here every workgroup
	// adds up ITS share of the (tiles touched, visible) pairs the projection kernel's waves left (a few dozen pairs: one
	// load per thread) and stores the partial sums; an extra workgroup of the row-prefix launch adds those up and stores the
	// totals into MAPPED HOST memory -- no memset, no atomics, no copy kernel in the stream, no launch of its own.  (One workgroup
	// summing all 31 k pairs of a 2 M-Gaussian model took 30 us: a chain of dependent round trips.)
	if (hc.pairs) {
		__shared__ unsigned long long s_t[SORT_THREADS / 64];
		__shared__ uint32_t s_v[SORT_THREADS / 64];
		const int share = (hc.n + nblocks - 1) / nblocks;
		__shared__ uint32_t s_m[SORT_THREADS / 64];
		unsigned long long t = 0ull;
		uint32_t v = 0u, mx = 0u;
		for (int i = (int)blockIdx.x * share + (int)threadIdx.x; i < min(hc.n, ((int)blockIdx.x + 1) * share); i += SORT_THREADS) {
			const uint4 c = hc.pairs[i];
			t += c.x;
			v += c.y;
			mx = max(mx, c.z);
		}
		const unsigned long long wt = wave_sum_u64_by_parts(t);
		const uint32_t wv = wave_sum_u32(v), wm = wave_max_u32(mx);
		if (lane_id() == 0) {
			s_t[wave_id()] = wt;
			s_v[wave_id()] = wv;
			s_m[wave_id()] = wm;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long T = 0ull;
			uint32_t V = 0u, M = 0u;
			for (int i = 0; i < SORT_THREADS / 64; i++) {
				T += s_t[i];
				V += s_v[i];
				M = max(M, s_m[i]);
			}
			hc.partials[blockIdx.x] = make_uint4((uint32_t)(T & 0xFFFFFFFFull), (uint32_t)(T >> 32), V, M);
		}
		__syncthreads();
	}
	// n_dev (nullable): the number of elements lives on the device (the compacted depth sort: set by the first pass's
	// scatter); blocks beyond it still write their (all-zero) histogram columns.  skip_invalid: keys equal to
	// RADIX_INVALID_KEY are no elements at all (culled Gaussians: never counted, never scattered).
	__shared__ uint32_t s_hist[BINS];
#pragma unroll
	for (int d = (int)threadIdx.x; d < BINS; d += SORT_THREADS) s_hist[d] = 0;
	if (n_dev) n = min(n, (int)*n_dev);
	__syncthreads();
	const uint32_t dmask = (1u << nbits) - 1u;
	const int wbase = blockIdx.x * SORT_CHUNK + wave_id() * SORT_ITEMS_PER_WAVE;
	// all 16 loads in flight before the first ballot (one HBM latency per wave instead of sixteen)
	uint32_t key[SORT_ROUNDS];
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const int i = wbase + r * 64 + lane_id();
		key[r] = i < n ? keys[i] : 0u;
	}
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const int i = wbase + r * 64 + lane_id();
		// one LDS atomic per key: even 64 lanes on one bin (64 serialised adds) cost less than the ~60 VALU of a ballot
		// match; only the scatter kernel needs the match, for its stable ranks
		if (i < n && !(skip_invalid && key[r] == RADIX_INVALID_KEY)) atomicAdd(&s_hist[((key[r] - bias) >> shift) & dmask], 1u);
	}
	__syncthreads();
	// (rows beyond the pass's digits -- 128 of the 256 in a 7-bit pass of the tile sort -- are never read: not written either)
	for (int d = (int)threadIdx.x; d < (1 << nbits); d += SORT_THREADS) hist[(size_t)d * nblocks + blockIdx.x] = s_hist[d];
}

// One workgroup per digit: exclusive scan of that digit's row hist[d][0..nblocks) in place, row total to
// totals[d].  Together with a 256-entry scan of the totals inside the scatter kernel this replaces the
// generic three-launch scan of the whole [256][nblocks] table (3 launches per pass instead of 5).
__global__ void __launch_bounds__(SCAN_THREADS)
radix_row_prefix_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, int nblocks, int nrows, const RadixHostCount hc)
{
	__shared__ uint32_t s_wave[4];
	if ((int)blockIdx.x == nrows) {
		// the extra workgroup of the depth sort's first pass (radix_hist_kernel): the histogram blocks' partial counts -> the totals,
		// stored into mapped host memory; the event the host waits for is recorded behind this launch
		__shared__ unsigned long long s_t[SCAN_THREADS / 64];
		__shared__ uint32_t s_m[SCAN_THREADS / 64];
		unsigned long long t = 0ull;
		uint32_t v = 0u, mx = 0u;
		for (int i = (int)threadIdx.x; i < nblocks; i += SCAN_THREADS) {
			const uint4 c = hc.partials[i];
			t += (unsigned long long)c.x | ((unsigned long long)c.y << 32);
			v += c.z;
			mx = max(mx, c.w);
		}
		const unsigned long long wt = wave_sum_u64_by_parts(t);
		const uint32_t wv = wave_sum_u32(v), wm = wave_max_u32(mx);
		if (lane_id() == 0) {
			s_t[wave_id()] = wt;
			s_wave[wave_id()] = wv;
			s_m[wave_id()] = wm;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned long long T = 0ull;
			uint32_t V = 0u, M = 0u;
			for (int i = 0; i < SCAN_THREADS / 64; i++) {
				T += s_t[i];
				V += s_wave[i];
				M = max(M, s_m[i]);
			}
			hc.host_out[0] = (uint32_t)(T & 0xFFFFFFFFull);
			hc.host_out[1] = (uint32_t)(T >> 32);
			hc.host_out[2] = V;
			hc.host_out[3] = M;
		}
		return;
	}
	uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
	uint32_t carry = 0;
	// (four consecutive entries per thread and trip: one block scan per 1 024 entries -- a row of the depth sort in one trip)
	constexpr int IPT = 4;
	for (int base = 0; base < nblocks; base += SCAN_THREADS * IPT) {
		const int i0 = base + (int)threadIdx.x * IPT;
		uint32_t v[IPT], sum = 0;
#pragma unroll
		for (int j = 0; j < IPT; j++) {
			v[j] = i0 + j < nblocks ? row[i0 + j] : 0u;
			sum += v[j];
		}
		uint32_t tot;
		uint32_t run = carry + block_excl_scan_256(sum, &tot, s_wave);
#pragma unroll
		for (int j = 0; j < IPT; j++) {
			if (i0 + j < nblocks) row[i0 + j] = run;
			run += v[j];
		}
		carry += tot;
	}
	if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int BINS>
__global__ void __launch_bounds__(SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n, int shift, int nbits,
                     const uint32_t* __restrict__ hist_rows, const uint32_t* __restrict__ totals, int nblocks,
                     const uint32_t* __restrict__ n_dev, int skip_invalid, uint32_t* __restrict__ count_out, uint32_t bias,
                     uint2* __restrict__ ranges_out)
{
	constexpr int DPT = BINS / SORT_THREADS;   // digits per thread in phase B: thread t owns digits t DPT .. t DPT + DPT - 1
	if (n_dev) n = min(n, (int)*n_dev);
	__shared__ uint32_t s_whist[4][BINS];  // per-wave digit counts, then per-wave running write cursors
	__shared__ uint32_t s_gbase[BINS];     // global position of local element i of digit d = s_gbase[d] + i
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_keys[SORT_CHUNK];
	__shared__ uint32_t s_vals[SORT_CHUNK];

	const int w = wave_id(), l = lane_id();
	const int tid = (int)threadIdx.x;
	const uint32_t dmask = (1u << nbits) - 1u;
	const int cbase = blockIdx.x * SORT_CHUNK;
	const int wbase = cbase + w * SORT_ITEMS_PER_WAVE;

#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int d = tid; d < BINS; d += SORT_THREADS) s_whist[i][d] = 0;
	__syncthreads();

	uint32_t key[SORT_ROUNDS], val[SORT_ROUNDS];
	uint32_t place[SORT_ROUNDS];   // rank inside the lane's digit group of this round (low byte) | group size << 8
	// Phase A: load (coalesced 256 B per wave instruction) and count digits per wave.
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const int i = wbase + r * 64 + l;
		const bool valid = i < n;
		key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
		val[r] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
	}
	uint32_t live = 0xFFFFu;   // bit r: element r of this lane exists (in range and, when skip_invalid, not the invalid key)
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const int i = wbase + r * 64 + l;
		const bool valid = i < n && !(skip_invalid && key[r] == RADIX_INVALID_KEY);
		if (!valid) live &= ~(1u << r);
		const uint32_t d = ((key[r] - bias) >> shift) & dmask;
		const unsigned long long m = wave_match_digit(d, nbits, valid);
		const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt()), size = (uint32_t)__popcll(m);
		place[r] = rank | (size << 8);   // the ballots are not repeated in phase C
		// the lowest lane of each digit group adds the group size; groups of one wave touch distinct bins
		if (valid && rank == 0u) s_whist[w][d] += size;
		wave_fence();
	}
	__syncthreads();

	// Phase B: thread d owns digit d.  Local layout of the chunk = digits ascending, inside a digit
	// waves ascending, inside a wave original order.
	uint32_t block_total;   // elements of this chunk that exist (the same value in every thread)
	{
		uint32_t c[DPT][4], tot[DPT], sum = 0, gsum = 0, gt[DPT];
#pragma unroll
		for (int j = 0; j < DPT; j++) {
			const int d = tid * DPT + j;
#pragma unroll
			for (int w4 = 0; w4 < 4; w4++) c[j][w4] = s_whist[w4][d];
			tot[j] = c[j][0] + c[j][1] + c[j][2] + c[j][3];
			sum += tot[j];
			// (the rows of the digits this pass does not have are neither written nor scanned)
			gt[j] = d < (1 << nbits) ? totals[d] : 0u;
			gsum += gt[j];
		}
		uint32_t lstart = block_excl_scan_256(sum, &block_total, s_wave);
		uint32_t all;
		uint32_t digit_base = block_excl_scan_256(gsum, &all, s_wave);   // elements with a smaller digit
#pragma unroll
		for (int j = 0; j < DPT; j++) {
			const int d = tid * DPT + j;
			s_whist[0][d] = lstart;
			s_whist[1][d] = lstart + c[j][0];
			s_whist[2][d] = lstart + c[j][0] + c[j][1];
			s_whist[3][d] = lstart + c[j][0] + c[j][1] + c[j][2];
			s_gbase[d] = digit_base + (d < (1 << nbits) ? hist_rows[(size_t)d * nblocks + blockIdx.x] : 0u) - lstart;
			// (a one-pass sort on the whole key: the run of key d in the output is [digit_base, digit_base + its total))
			if (ranges_out && blockIdx.x == 0 && gt[j] != 0u) ranges_out[d] = make_uint2(digit_base, digit_base + gt[j]);
			lstart += tot[j];
			digit_base += gt[j];
		}
		if (count_out && blockIdx.x == 0 && tid == 0) *count_out = all;   // the elements that exist: later passes run over them only
	}
	__syncthreads();
	// (block_total is the same in every thread: block_excl_scan_256 returns the block sum to all)

	// Phase C: stable placement into the LDS tile.
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const bool valid = (live >> r) & 1u;
		const uint32_t d = ((key[r] - bias) >> shift) & dmask;
		const uint32_t rank = place[r] & 0xFFu, size = place[r] >> 8;
		uint32_t cursor = 0;
		if (valid) cursor = s_whist[w][d];
		wave_fence();  // every lane has read the cursor before the group leader advances it
		if (valid) {
			s_keys[cursor + rank] = key[r];
			s_vals[cursor + rank] = val[r];
			if (rank == 0) s_whist[w][d] = cursor + size;
		}
		wave_fence();
	}
	__syncthreads();

	// Phase D: write the locally sorted tile; lanes of a digit run hit consecutive addresses.
	const int count = (int)block_total;
	for (int i = tid; i < count; i += SORT_THREADS) {
		const uint32_t k = s_keys[i];
		const uint32_t d = ((k - bias) >> shift) & dmask;
		const uint32_t pos = s_gbase[d] + (uint32_t)i;
		keys_out[pos] = k;
		vals_out[pos] = s_vals[i];
	}
}

int launch_radix_sort(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_ping, uint32_t* vals_ping,
                      uint32_t* keys_pong, uint32_t* vals_pong, int n, int begin_bit, int end_bit,
                      uint32_t* scratch, hipStream_t stream, uint32_t** keys_res, uint32_t** vals_res, uint32_t* compact_count,
                      const RadixHostCount* host_count, bool first_hist_ready, int digit_bits, uint32_t bias, uint2* ranges_out)
{
	// compact_count (nullable, device word): keys equal to RADIX_INVALID_KEY are dropped by the first pass, which leaves the
	// number of remaining elements there; the later passes (and the caller's consumers) run over that many elements only.
	// digit_bits: 8 (RADIX_BITS) or 9 (RADIX_BITS_WIDE: 512-bin kernels; scratch of sort_scratch_elems_wide(n)); bias: subtracted
	// from every key in front of the digits (the sort is on key - bias, bits [begin_bit, end_bit): the caller guarantees that the
	// bits above end_bit of key - bias are zero for every element, or handles the exception itself -- gsr_forward's depth sort).
	if (digit_bits != RADIX_BITS && digit_bits != RADIX_BITS_WIDE && digit_bits != RADIX_BITS_ONE_PASS) return GSR_ERR_INVALID_ARG;
	const int passes = end_bit > begin_bit ? div_up(end_bit - begin_bit, digit_bits) : 0;
	if (ranges_out && (passes != 1 || begin_bit != 0)) return GSR_ERR_INVALID_ARG;
	*keys_res = (passes % 2) ? keys_pong : keys_ping;
	*vals_res = (passes % 2) ? vals_pong : vals_ping;
	if (n <= 0) return GSR_OK;
	const int nb = sort_blocks(n);
	const int hist_elems = (1 << digit_bits) * nb;
	uint32_t* hist = scratch;
	uint32_t* totals = scratch + hist_elems;
	if (passes == 0) {
		// degenerate: nothing to sort on; result must still be materialised in the ping buffers
		GSR_HIP(hipMemcpyAsync(keys_ping, keys_in, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
		if (vals_in) GSR_HIP(hipMemcpyAsync(vals_ping, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
		else return GSR_ERR_INVALID_ARG;
		return GSR_OK;
	}
	const uint32_t* kin = keys_in;
	const uint32_t* vin = vals_in;
	// the key bits are spread evenly over the passes (13 tile bits: 7 + 6, not 8 + 5): a pass scatters into 2^nbits streams,
	// and with fewer streams a workgroup's runs per stream are longer, i.e. its writes better coalesced
	const int bits_per_pass = div_up(end_bit - begin_bit, passes);
	const int bins = bits_per_pass <= RADIX_BITS ? RADIX_BINS : (bits_per_pass <= RADIX_BITS_WIDE ? 512 : 2048);
	for (int p = 0; p < passes; p++) {
		const int shift = begin_bit + p * bits_per_pass;
		const int nbits = min(bits_per_pass, end_bit - shift);
		uint32_t* kout = (p % 2 == 0) ? keys_pong : keys_ping;
		uint32_t* vout = (p % 2 == 0) ? vals_pong : vals_ping;
		const uint32_t* n_dev = (compact_count && p > 0) ? compact_count : nullptr;
		const int skip = (compact_count && p == 0) ? 1 : 0;
		// (the first pass may carry the forward pass's count for the host: RadixHostCount, state.h)
		const bool counts_ride = p == 0 && host_count && host_count->pairs;
		const RadixHostCount hc = counts_ride ? *host_count : RadixHostCount{};
		// (first_hist_ready: the producer of the keys has counted the first pass's digits into `hist` itself -- the instance emission)
		if (!(p == 0 && first_hist_ready)) {
			if (bins == 2048) GSR_LAUNCH(radix_hist_kernel<2048>, nb, SORT_THREADS, stream, kin, n, shift, nbits, hist, nb, n_dev, skip, hc, bias);
			else if (bins == 512) GSR_LAUNCH(radix_hist_kernel<512>, nb, SORT_THREADS, stream, kin, n, shift, nbits, hist, nb, n_dev, skip, hc, bias);
			else GSR_LAUNCH(radix_hist_kernel<RADIX_BINS>, nb, SORT_THREADS, stream, kin, n, shift, nbits, hist, nb, n_dev, skip, hc, bias);
		}
		GSR_LAUNCH(radix_row_prefix_kernel, (1 << nbits) + (counts_ride ? 1 : 0), SCAN_THREADS, stream, hist, totals, nb, 1 << nbits, hc);
		if (counts_ride && hc.ready) GSR_HIP(hipEventRecord((hipEvent_t)hc.ready, stream));
		if (bins == 2048)
			GSR_LAUNCH(radix_scatter_kernel<2048>, nb, SORT_THREADS, stream, kin, vin, kout, vout, n, shift, nbits, (const uint32_t*)hist,
			           (const uint32_t*)totals, nb, n_dev, skip, skip ? compact_count : (uint32_t*)nullptr, bias, ranges_out);
		else if (bins == 512)
			GSR_LAUNCH(radix_scatter_kernel<512>, nb, SORT_THREADS, stream, kin, vin, kout, vout, n, shift, nbits, (const uint32_t*)hist,
			           (const uint32_t*)totals, nb, n_dev, skip, skip ? compact_count : (uint32_t*)nullptr, bias, ranges_out);
		else
			GSR_LAUNCH(radix_scatter_kernel<RADIX_BINS>, nb, SORT_THREADS, stream, kin, vin, kout, vout, n, shift, nbits, (const uint32_t*)hist,
			           (const uint32_t*)totals, nb, n_dev, skip, skip ? compact_count : (uint32_t*)nullptr, bias, ranges_out);
		kin = kout;
		vin = vout;
	}
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
