// tile_depth_sort.hip -- the depth order of every tile's list, established per tile (tile-first binning, gsr_api.hip).
//
// The reference sorts all R instances on the 64-bit key tile << 32 | depth bits (rasterizer_impl.cu:70-111, 303-308).  The
// depth-first arrangement of this library sorts the GAUSSIANS by depth first (nine launches, whatever the size) and lets the
// stable tile sort carry that order into the tiles.  Tile-first: the visible Gaussians are only compacted (ascending id:
// sort.hip, compact_scan_kernel), their instances reach the tiles in id order, and one workgroup per tile sorts the tile's list
// by the depth bits with a stable LSD radix sort -- in LDS when the list has up to 2 048 entries, chunk by chunk through the
// tile's own segments of the binning buffer's free arrays when it is longer.  Equal depths keep ascending id, the list order of
// the reference: the final (tile, depth bits, id) order is the same, bit for bit.
//
// Per pass the digits are ranked as in radix_scatter_kernel (sort.hip): per wave by ballots (wave_match_digit), across the
// four waves and the digits by one block scan; only the bits in which the tile's keys differ are sorted (key - min over the
// list: a tile that sees one wall sorts a dozen bits).
#include "state.h"
#include "wave64.h"
#include "kernels.h"

namespace gsr {

constexpr int TDS_THREADS = 256, TDS_WAVES = TDS_THREADS / 64;
constexpr int TDS_CHUNK = 2048;                       // entries ranked at a time: eight per thread
constexpr int TDS_ROUNDS = TDS_CHUNK / TDS_THREADS;   // rounds of 64 lane-consecutive entries per wave
constexpr int TDS_BITS = 8, TDS_BINS = 1 << TDS_BITS;
static_assert(TDS_BINS == TDS_THREADS, "thread d owns digit d");

struct TdsShared {
	uint32_t whist[TDS_WAVES][TDS_BINS];   // per-wave digit counts, then per-wave write cursors
	uint32_t gbase[TDS_BINS];              // long lists: destination of local element i of digit d = gbase[d] + i
	uint32_t cur[TDS_BINS];                // long lists: the pass's running cursor per digit
	uint32_t keys[TDS_CHUNK];
	uint32_t vals[TDS_CHUNK];
	uint32_t wave[4];
	uint32_t red[2 * TDS_WAVES];
};

// Ranks the chunk the workgroup holds in registers -- wave w, round r, lane l holds local element w * span + r * 64 + l, `live`
// bit r says whether it exists -- by the digit ((key - bias) >> shift) & mask, stable, and leaves it in S.keys / S.vals in that
// order.  between(lstart, tot): called by thread d with digit d's first local position and count, after the scan and in front of
// the barrier that precedes the placement.
template <class Between>
__device__ __forceinline__ void tds_rank_chunk(const uint32_t (&key)[TDS_ROUNDS], const uint32_t (&val)[TDS_ROUNDS], uint32_t live, int rounds,
                                               uint32_t bias, int shift, int nbits, TdsShared& S, Between&& between)
{
	const int w = wave_id(), tid = (int)threadIdx.x;
	const uint32_t dmask = (1u << nbits) - 1u;
#pragma unroll
	for (int i = 0; i < TDS_WAVES; i++) S.whist[i][tid] = 0u;
	__syncthreads();
	uint32_t place[TDS_ROUNDS];   // rank inside the lane's digit group of the round (low byte) | group size << 8
#pragma unroll
	for (int r = 0; r < TDS_ROUNDS; r++) {
		place[r] = 0u;
		if (r < rounds) {   // (block-uniform)
			const bool valid = (live >> r) & 1u;
			const uint32_t d = ((key[r] - bias) >> shift) & dmask;
			const unsigned long long m = wave_match_digit(d, nbits, valid);
			const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt()), size = (uint32_t)__popcll(m);
			place[r] = rank | (size << 8);
			if (valid && rank == 0u) S.whist[w][d] += size;   // (the groups of one wave touch distinct bins)
			wave_fence();
		}
	}
	__syncthreads();
	{
		const uint32_t c0 = S.whist[0][tid], c1 = S.whist[1][tid], c2 = S.whist[2][tid], c3 = S.whist[3][tid];
		const uint32_t tot = c0 + c1 + c2 + c3;
		uint32_t all;
		const uint32_t lstart = block_excl_scan_256(tot, &all, S.wave);
		S.whist[0][tid] = lstart;
		S.whist[1][tid] = lstart + c0;
		S.whist[2][tid] = lstart + c0 + c1;
		S.whist[3][tid] = lstart + c0 + c1 + c2;
		between(lstart, tot);
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < TDS_ROUNDS; r++) {
		if (r < rounds) {
			const bool valid = (live >> r) & 1u;
			const uint32_t d = ((key[r] - bias) >> shift) & dmask;
			const uint32_t rank = place[r] & 0xFFu, size = place[r] >> 8;
			uint32_t cursor = 0;
			if (valid) cursor = S.whist[w][d];
			wave_fence();   // every lane has read the cursor before the group leader advances it
			if (valid) {
				S.keys[cursor + rank] = key[r];
				S.vals[cursor + rank] = val[r];
				if (rank == 0u) S.whist[w][d] = cursor + size;
			}
			wave_fence();
		}
	}
	__syncthreads();
}

// the smallest and the largest key of the workgroup's elements (the same values in every thread)
__device__ __forceinline__ void tds_block_min_max(uint32_t lo, uint32_t hi, TdsShared& S, uint32_t& kmin, uint32_t& kmax)
{
	const uint32_t wlo = ~wave_max_u32(~lo), whi = wave_max_u32(hi);
	__syncthreads();
	if (lane_id() == 0) {
		S.red[wave_id()] = wlo;
		S.red[TDS_WAVES + wave_id()] = whi;
	}
	__syncthreads();
	kmin = min(min(S.red[0], S.red[1]), min(S.red[2], S.red[3]));
	kmax = max(max(S.red[TDS_WAVES], S.red[TDS_WAVES + 1]), max(S.red[TDS_WAVES + 2], S.red[TDS_WAVES + 3]));
}

// xk / xv: the tile sort's spare (key, value) arrays; yk: one more array of R words (the forward blend's flag planes, not yet
// written).  Tile t only touches [ranges[t].x, ranges[t].y) of each.
__global__ void __launch_bounds__(TDS_THREADS)
tile_depth_sort_kernel(const uint2* __restrict__ ranges, int tiles, const uint32_t* __restrict__ depth_key, uint32_t* point_list,
                       uint32_t* xk, uint32_t* xv, uint32_t* yk)
{
	__shared__ TdsShared S;
	const int tile = (int)blockIdx.x;
	if (tile >= tiles) return;
	const uint2 range = ranges[tile];
	const int n = (int)(range.y - range.x);
	if (n <= 1) return;
	const int w = wave_id(), l = lane_id(), tid = (int)threadIdx.x;
	uint32_t* const pl = point_list + range.x;
	uint32_t key[TDS_ROUNDS], val[TDS_ROUNDS];

	if (n <= TDS_CHUNK) {
		// ---- the list fits the workgroup's registers and LDS: every pass stays there
		const int span = ((n + TDS_WAVES - 1) / TDS_WAVES + 63) & ~63;   // elements per wave, a multiple of 64 (<= 512)
		const int rounds = span >> 6;
		uint32_t live = 0u, lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
		for (int r = 0; r < TDS_ROUNDS; r++) {
			const int e = w * span + r * 64 + l;
			key[r] = 0u;
			val[r] = 0u;
			if (r < rounds && e < n) {
				val[r] = pl[e];
				live |= 1u << r;
			}
		}
#pragma unroll
		for (int r = 0; r < TDS_ROUNDS; r++)
			if ((live >> r) & 1u) {
				key[r] = depth_key[val[r]];
				lo = min(lo, key[r]);
				hi = max(hi, key[r]);
			}
		uint32_t kmin, kmax;
		tds_block_min_max(lo, hi, S, kmin, kmax);
		if (kmax == kmin) return;   // (one depth: the id order the list arrived in is the answer)
		const int bits = 32 - __builtin_clz(kmax - kmin);
		const int passes = (bits + TDS_BITS - 1) / TDS_BITS, bpp = (bits + passes - 1) / passes;
		for (int p = 0; p < passes; p++) {
			const int shift = p * bpp, nbits = min(bpp, bits - shift);
			tds_rank_chunk(key, val, live, rounds, kmin, shift, nbits, S, [](uint32_t, uint32_t) {});
			if (p + 1 < passes) {
#pragma unroll
				for (int r = 0; r < TDS_ROUNDS; r++)
					if ((live >> r) & 1u) {
						const int e = w * span + r * 64 + l;
						key[r] = S.keys[e];
						val[r] = S.vals[e];
					}
				// (the next pass writes S.keys only behind two more barriers)
			}
		}
		for (int i = tid; i < n; i += TDS_THREADS) pl[i] = S.vals[i];
		return;
	}

	// ---- a long list: the same passes chunk by chunk, ping-pong between (yk, point_list) and (xk, xv) -- the tile's own segments
	uint32_t* ak = yk + range.x;
	uint32_t* av = pl;
	uint32_t* bk = xk + range.x;
	uint32_t* bv = xv + range.x;
	uint32_t lo = 0xFFFFFFFFu, hi = 0u;
	for (int i = tid; i < n; i += TDS_THREADS) {
		const uint32_t k = depth_key[pl[i]];
		ak[i] = k;
		lo = min(lo, k);
		hi = max(hi, k);
	}
	uint32_t kmin, kmax;
	tds_block_min_max(lo, hi, S, kmin, kmax);   // (its barriers also make the keys just written visible to the whole workgroup)
	if (kmax == kmin) return;
	const int bits = 32 - __builtin_clz(kmax - kmin);
	const int passes = (bits + TDS_BITS - 1) / TDS_BITS, bpp = (bits + passes - 1) / passes;
	for (int p = 0; p < passes; p++) {
		const int shift = p * bpp, nbits = min(bpp, bits - shift);
		const uint32_t dmask = (1u << nbits) - 1u;
		// the pass's digit counts over the whole list -> the first position of every digit
		S.cur[tid] = 0u;
		__syncthreads();
		for (int i = tid; i < n; i += TDS_THREADS) atomicAdd(&S.cur[((ak[i] - kmin) >> shift) & dmask], 1u);
		__syncthreads();
		{
			const uint32_t c = S.cur[tid];
			uint32_t all;
			const uint32_t ex = block_excl_scan_256(c, &all, S.wave);
			S.cur[tid] = ex;
		}
		__syncthreads();
		for (int base = 0; base < n; base += TDS_CHUNK) {
			const int cn = min(TDS_CHUNK, n - base);
			uint32_t live = 0u;
#pragma unroll
			for (int r = 0; r < TDS_ROUNDS; r++) {
				const int e = w * (TDS_CHUNK / TDS_WAVES) + r * 64 + l;
				key[r] = 0u;
				val[r] = 0u;
				if (e < cn) {
					key[r] = ak[base + e];
					val[r] = av[base + e];
					live |= 1u << r;
				}
			}
			tds_rank_chunk(key, val, live, TDS_ROUNDS, kmin, shift, nbits, S, [&](uint32_t lstart, uint32_t tot) {
				S.gbase[tid] = S.cur[tid] - lstart;
				S.cur[tid] += tot;
			});
			for (int i = tid; i < cn; i += TDS_THREADS) {
				const uint32_t k = S.keys[i];
				const uint32_t pos = S.gbase[((k - kmin) >> shift) & dmask] + (uint32_t)i;
				bk[pos] = k;
				bv[pos] = S.vals[i];
			}
			// (the next chunk writes S.gbase / S.keys only behind the barriers of its own ranking)
		}
		__syncthreads();   // the pass's output is complete (and visible to the workgroup) before it is read as the next input
		uint32_t* t = ak; ak = bk; bk = t;
		t = av; av = bv; bv = t;
	}
	if (av != pl)   // an odd number of passes ended in the spare arrays
		for (int i = tid; i < n; i += TDS_THREADS) pl[i] = av[i];
}

int launch_tile_depth_sort(const uint2* ranges, int tiles, const uint32_t* depth_key, uint32_t* point_list, uint32_t* spare_keys,
                           uint32_t* spare_vals, uint32_t* spare_words, hipStream_t stream)
{
	if (!ranges || !depth_key || !point_list || !spare_keys || !spare_vals || !spare_words) return GSR_ERR_INVALID_ARG;
	if (tiles <= 0) return GSR_OK;
	GSR_LAUNCH(tile_depth_sort_kernel, tiles, TDS_THREADS, stream, ranges, tiles, (const uint32_t*)depth_key, point_list, spare_keys, spare_vals, spare_words);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // namespace gsr
