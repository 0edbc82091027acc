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
// Block-wide exclusive scan of one value per thread (256 threads = 4 waves).
// Returns the exclusive prefix; *total receives the block sum (valid in all threads).
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total, uint32_t* s_wave /*[4]*/)
{
	const uint32_t incl = wave_incl_scan_u32(v);
	const int w = wave_id(), l = lane_id();
	__syncthreads();  // s_wave reuse across calls
	if (l == 63) s_wave[w] = incl;
	__syncthreads();
	uint32_t base = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const uint32_t sw = s_wave[i];
		if (i < w) base += sw;
		tot += sw;
	}
	*total = tot;
	return base + incl - v;
}

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
                        const uint32_t* __restrict__ n_dev)
{
	__shared__ uint32_t s_wave[4];
	const int base = blockIdx.x * items_per_block;
	const int end_all = min(n, base + items_per_block);
	const int end = n_dev ? min(end_all, (int)*n_dev) : end_all;
	uint32_t acc = 0;
	if (n_dev)
		for (int i = max(base, end) + (int)threadIdx.x; i < end_all; i += SCAN_THREADS) staged[i] = 0u;
	for (int i = base + (int)threadIdx.x; i < end; i += SCAN_THREADS) {
		const uint2 r = rect[gather[i]];
		const uint32_t v = ((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16));
		rect_sorted[i] = r;
		staged[i] = v;
		acc += v;
	}
	uint32_t tot;
	block_excl_scan_256(acc, &tot, s_wave);
	if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_apply_kernel(const uint32_t* in, const uint32_t* __restrict__ gather, uint32_t* out,   // in may alias out (staged input)
                  const uint32_t* __restrict__ block_sums, int n, int items_per_block, int inclusive)
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
	for (int b = base; b < end; b += SCAN_THREADS) {
		const int i = b + (int)threadIdx.x;
		const uint32_t v = i < end ? (gather ? in[gather[i]] : in[i]) : 0u;
		uint32_t tot;
		const uint32_t ex = block_excl_scan_256(v, &tot, s_wave);
		if (i < end) out[i] = carry + ex + (inclusive ? v : 0u);
		carry += tot;
	}
}

int launch_scan_rect_tiles(const uint2* rect, const uint32_t* gather, uint32_t* out, uint2* rect_sorted, int n, uint32_t* scratch,
                           hipStream_t stream, const uint32_t* n_dev)
{
	if (!rect || !gather || !out || !rect_sorted) return GSR_ERR_INVALID_ARG;
	if (n <= 0) return GSR_OK;
	const int ipb = scan_items_per_block(n);
	const int nb = div_up(n, ipb);
	GSR_LAUNCH(scan_reduce_rect_kernel, nb, SCAN_THREADS, stream, rect, gather, out, rect_sorted, scratch, n, ipb, n_dev);
	GSR_LAUNCH(scan_apply_kernel, nb, SCAN_THREADS, stream, (const uint32_t*)out, (const uint32_t*)nullptr, out,
	           (const uint32_t*)scratch, n, ipb, 0);
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
	           (const uint32_t*)scratch, n, ipb, inclusive ? 1 : 0);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

// ------------------------------------------------------------------ radix sort
// Pass structure (per digit of up to 8 bits): histogram -> scatter -- TWO launches (round 3: three; round 1: five).
// Block b always owns elements [b*SORT_CHUNK, (b+1)*SORT_CHUNK); wave w of the block owns the SORT_ITEMS_PER_WAVE-element
// sub-range starting at w*SORT_ITEMS_PER_WAVE and walks it in rounds of 64 lane-consecutive elements, so the original order
// inside a digit is (wave, round, lane).
//
// Where block b's elements of digit d go = (elements with a smaller digit) + (elements of digit d in the blocks before b).
// Until round 3 a kernel of its own scanned every digit's row of a [bin][block] table between the two launches
// (radix_row_prefix: 5.8 us per pass for 1 MB of table -- a launch slot, not work).  Now the blocks form GROUPS of SORT_GROUP:
// the histogram kernel leaves its counts as the row hist[b][.] (ONE contiguous store; the [bin][block] layout cost 256 scattered
// words per block) and adds them to the group's row gsum[b / SORT_GROUP][.] (one atomic per non-empty digit: integers, so the
// result does not depend on the order).  The block that finishes LAST (a ticket counter behind a release fence -- no block
// ever waits for another) turns the group rows into their exclusive prefix over the groups, in place, and leaves the digits'
// bases dbase[.]; the scatter kernel's thread d then needs gsum[its group][d], dbase[d] and the hist rows of the blocks in
// front of it inside its group (<= 31, coalesced over d).  (First form of this round, measured: every scatter block summing
// all group rows itself -- 3 046 blocks x 127 rows x 512 B = 200 MB of L2 reads per pass over the instances, +8 us per
// scatter launch, a net loss against the row-prefix kernel it replaced.)
// gsum and the ticket have to be zero before the histogram kernel runs: two tables alternate between the passes, pass p's
// histogram kernel clears the one pass p + 1 will use, the last block resets the ticket, and the first pass's table is cleared
// by the caller (the forward pass's one memset covers the depth sort's) or by a memset here.

__global__ void __launch_bounds__(SORT_THREADS)
radix_hist_kernel(const uint32_t* __restrict__ keys, int n, int shift, int nbits, uint32_t* __restrict__ hist,
                  uint32_t* gsum, uint32_t* __restrict__ gsum_next, uint32_t* __restrict__ dbase, uint32_t* ticket, int ngroups,
                  const uint32_t* __restrict__ n_dev, int skip_invalid)
{
	// n_dev (nullable): the number of elements lives on the device (the compacted depth sort: set by the first pass's
	// scatter); blocks beyond it still write their (all-zero) histogram rows.  skip_invalid: keys equal to
	// RADIX_INVALID_KEY are no elements at all (culled Gaussians: never counted, never scattered).
	__shared__ uint32_t s_hist[RADIX_BINS];
	s_hist[threadIdx.x] = 0;
	if (gsum_next && (int)blockIdx.x < ngroups) gsum_next[(size_t)blockIdx.x * RADIX_BINS + threadIdx.x] = 0u;   // for the next pass
	if (n_dev) n = min(n, (int)*n_dev);
	__syncthreads();
	const uint32_t dmask = (1u << nbits) - 1u;
	const int wbase = blockIdx.x * SORT_CHUNK + wave_id() * SORT_ITEMS_PER_WAVE;
	// all loads in flight before the first atomic (one HBM latency per wave instead of SORT_ROUNDS)
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
		if (i < n && !(skip_invalid && key[r] == RADIX_INVALID_KEY)) atomicAdd(&s_hist[(key[r] >> shift) & dmask], 1u);
	}
	__syncthreads();
	// (digits beyond the pass's -- 128 of the 256 in a 7-bit pass of the tile sort -- are never read: not written either)
	const bool used = (int)threadIdx.x < (1 << nbits);
	if (used) {
		const uint32_t c = s_hist[threadIdx.x];
		hist[(size_t)blockIdx.x * RADIX_BINS + threadIdx.x] = c;
		if (c) atomicAdd(&gsum[(size_t)(blockIdx.x / SORT_GROUP) * RADIX_BINS + threadIdx.x], c);
	}
	// The last block to get here finishes the group table (threadFenceReduction pattern: every thread's stores and atomics are
	// released at agent scope before the block takes its ticket; the block that draws the last ticket acquires and sees them all)
	__shared__ uint32_t s_last;
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == (uint32_t)gridDim.x - 1u ? 1u : 0u;
	__syncthreads();
	if (!s_last) return;
	__threadfence();
	uint32_t run = 0;
	if (used) {
#pragma unroll 4
		for (int gq = 0; gq < ngroups; gq++) {
			uint32_t* p = gsum + (size_t)gq * RADIX_BINS + threadIdx.x;
			const uint32_t v = __atomic_load_n(p, __ATOMIC_RELAXED);   // (the adds were atomics: read them at the same scope)
			*p = run;                                                  // exclusive over the groups
			run += v;
		}
	}
	__shared__ uint32_t s_wave[4];
	uint32_t all;
	const uint32_t base = block_excl_scan_256(run, &all, s_wave);       // elements with a smaller digit
	dbase[threadIdx.x] = base;
	if (threadIdx.x == 0) {
		dbase[RADIX_BINS] = all;   // the elements that exist
		*ticket = 0u;              // ready for the next pass
	}
}

__global__ void __launch_bounds__(SORT_THREADS)
radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n, int shift, int nbits,
                     const uint32_t* __restrict__ hist, const uint32_t* __restrict__ gsum, const uint32_t* __restrict__ dbase,
                     const uint32_t* __restrict__ n_dev, int skip_invalid, uint32_t* __restrict__ count_out)
{
	if (n_dev) n = min(n, (int)*n_dev);
	__shared__ uint32_t s_whist[4][RADIX_BINS];  // per-wave digit counts, then per-wave running write cursors
	__shared__ uint32_t s_gbase[RADIX_BINS];     // global position of local element i of digit d = s_gbase[d] + i
	__shared__ uint32_t s_wave[4];
	__shared__ uint32_t s_keys[SORT_CHUNK];
	__shared__ uint32_t s_vals[SORT_CHUNK];

	const int w = wave_id(), l = lane_id();
	const int tid = (int)threadIdx.x;
	const uint32_t dmask = (1u << nbits) - 1u;
	const int cbase = blockIdx.x * SORT_CHUNK;
	const int wbase = cbase + w * SORT_ITEMS_PER_WAVE;

#pragma unroll
	for (int i = 0; i < 4; i++) s_whist[i][tid] = 0;
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
	// Digit tid's place among the blocks, from the two-level tables (header): the loads are independent of the keys and go
	// out right behind them.  before = elements of the digit in the groups in front of this block's, within = in the blocks in
	// front of it inside its group, base = elements with a smaller digit.
	const bool used = tid < (1 << nbits);   // (the rows of the digits this pass does not have are neither written nor read)
	uint32_t before = 0, within = 0, base = 0;
	if (used) {
		const int grp = (int)blockIdx.x / SORT_GROUP;
		before = gsum[(size_t)grp * RADIX_BINS + tid];
		base = dbase[tid];
#pragma unroll 8
		for (int b = grp * SORT_GROUP; b < (int)blockIdx.x; b++) within += hist[(size_t)b * RADIX_BINS + tid];
	}
	uint32_t live = (1u << SORT_ROUNDS) - 1u;   // bit r: element r of this lane exists (in range and, when skip_invalid, not the invalid key)
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const int i = wbase + r * 64 + l;
		const bool valid = i < n && !(skip_invalid && key[r] == RADIX_INVALID_KEY);
		if (!valid) live &= ~(1u << r);
		const uint32_t d = (key[r] >> shift) & dmask;
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
		const uint32_t c0 = s_whist[0][tid], c1 = s_whist[1][tid], c2 = s_whist[2][tid], c3 = s_whist[3][tid];
		const uint32_t tot = c0 + c1 + c2 + c3;
		const uint32_t lstart = block_excl_scan_256(tot, &block_total, s_wave);
		s_whist[0][tid] = lstart;
		s_whist[1][tid] = lstart + c0;
		s_whist[2][tid] = lstart + c0 + c1;
		s_whist[3][tid] = lstart + c0 + c1 + c2;
		s_gbase[tid] = base + before + within - lstart;
		if (count_out && blockIdx.x == 0 && tid == 0) *count_out = dbase[RADIX_BINS];   // the elements that exist: later passes run over them only
	}
	__syncthreads();
	// (block_total is the same in every thread: block_excl_scan_256 returns the block sum to all)

	// Phase C: stable placement into the LDS tile.
#pragma unroll
	for (int r = 0; r < SORT_ROUNDS; r++) {
		const bool valid = (live >> r) & 1u;
		const uint32_t d = (key[r] >> shift) & dmask;
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
		const uint32_t d = (k >> shift) & dmask;
		const uint32_t pos = s_gbase[d] + (uint32_t)i;
		keys_out[pos] = k;
		vals_out[pos] = s_vals[i];
	}
}

int launch_radix_sort(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_ping, uint32_t* vals_ping,
                      uint32_t* keys_pong, uint32_t* vals_pong, int n, int begin_bit, int end_bit,
                      uint32_t* scratch, hipStream_t stream, uint32_t** keys_res, uint32_t** vals_res, uint32_t* compact_count,
                      uint32_t* gsum0_zeroed)
{
	// compact_count (nullable, device word): keys equal to RADIX_INVALID_KEY are dropped by the first pass, which leaves the
	// number of remaining elements there; the later passes (and the caller's consumers) run over that many elements only.
	// gsum0_zeroed (nullable): sort_gsum_elems(n) words the caller has ALREADY zeroed on this stream (the first pass's group
	// table); without it the table lives in `scratch` and is cleared here (one memset more in the stream).
	const int passes = end_bit > begin_bit ? div_up(end_bit - begin_bit, RADIX_BITS) : 0;
	*keys_res = (passes % 2) ? keys_pong : keys_ping;
	*vals_res = (passes % 2) ? vals_pong : vals_ping;
	if (n <= 0) return GSR_OK;
	const int nb = sort_blocks(n);
	const int ngroups = div_up(nb, SORT_GROUP);
	uint32_t* hist = scratch;                                              // [nb][RADIX_BINS]
	uint32_t* gsum[2] = {scratch + (size_t)RADIX_BINS * nb,                // [ngroups][RADIX_BINS] + the ticket, even passes
	                     scratch + (size_t)RADIX_BINS * nb + sort_gsum_elems(n)};   // odd passes
	uint32_t* dbase = scratch + (size_t)RADIX_BINS * nb + 2 * sort_gsum_elems(n);   // [RADIX_BINS + 1]
	if (passes == 0) {
		// degenerate: nothing to sort on; result must still be materialised in the ping buffers
		GSR_HIP(hipMemcpyAsync(keys_ping, keys_in, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
		if (vals_in) GSR_HIP(hipMemcpyAsync(vals_ping, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
		else return GSR_ERR_INVALID_ARG;
		return GSR_OK;
	}
	if (gsum0_zeroed) gsum[0] = gsum0_zeroed;
	else GSR_HIP(hipMemsetAsync(gsum[0], 0, sort_gsum_elems(n) * sizeof(uint32_t), stream));
	uint32_t* ticket = gsum[0] + (size_t)RADIX_BINS * ngroups;   // (behind the first table: zeroed with it, reset by its last user)
	const uint32_t* kin = keys_in;
	const uint32_t* vin = vals_in;
	// the key bits are spread evenly over the passes (13 tile bits: 7 + 6, not 8 + 5): a pass scatters into 2^nbits streams,
	// and with fewer streams a workgroup's runs per stream are longer, i.e. its writes better coalesced
	const int bits_per_pass = div_up(end_bit - begin_bit, passes);
	for (int p = 0; p < passes; p++) {
		const int shift = begin_bit + p * bits_per_pass;
		const int nbits = min(bits_per_pass, end_bit - shift);
		uint32_t* kout = (p % 2 == 0) ? keys_pong : keys_ping;
		uint32_t* vout = (p % 2 == 0) ? vals_pong : vals_ping;
		const uint32_t* n_dev = (compact_count && p > 0) ? compact_count : nullptr;
		const int skip = (compact_count && p == 0) ? 1 : 0;
		uint32_t* g_now = gsum[p & 1];
		uint32_t* g_next = p + 1 < passes ? gsum[(p + 1) & 1] : nullptr;
		GSR_LAUNCH(radix_hist_kernel, nb, SORT_THREADS, stream, kin, n, shift, nbits, hist, g_now, g_next, dbase, ticket, ngroups,
		           n_dev, skip);
		GSR_LAUNCH(radix_scatter_kernel, nb, SORT_THREADS, stream, kin, vin, kout, vout, n, shift, nbits,
		           (const uint32_t*)hist, (const uint32_t*)g_now, (const uint32_t*)dbase, n_dev, skip,
		           skip ? compact_count : (uint32_t*)nullptr);
		kin = kout;
		vin = vout;
	}
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
