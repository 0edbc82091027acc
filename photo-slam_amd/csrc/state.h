// state.h -- HBM layout of the three caller-owned scratch buffers and of the per-Gaussian
// blend record.  This replaces GeometryState / BinningState / ImageState of the reference
// (cuda_rasterizer/rasterizer_impl.h:32-62); callers treat the buffers as opaque bytes.
#pragma once
#include "rt.h"

namespace gsr {

constexpr int TILE = 16;              // BLOCK_X == BLOCK_Y, cuda_rasterizer/config.h:16-17 (part of the parity contract)
constexpr int TILE_PIXELS = TILE * TILE;
constexpr uint32_t DEPTH_KEY_CULLED = 0xFFFFFFFFu;
constexpr uint32_t DEPTH_KEY_BIAS = 0x3E4CCCCDu;   // bits(0.2f): a visible Gaussian's key is larger (preprocess.hip: vz <= 0.2f is culled)
// The forward preprocess runs PRE_THREADS / 64 waves per workgroup; every wave leaves its (tiles touched, visible) pair with a
// plain store -- no atomics, so no array that would have to be zeroed first (until round 5: atomics spread over 1024 words behind
// a memset, and a 4 KiB copy to the host: two runtime blit kernels with their bubbles in front of and in the middle of the forward pass)
constexpr int PRE_THREADS = 128;
static inline size_t wave_count_slots(size_t P) { return ((P + PRE_THREADS - 1) / PRE_THREADS) * (PRE_THREADS / 64); }
constexpr int HOST_COUNT_WORDS = 4;   // what the forward pass hands to the host: [0] tiles lo, [1] tiles hi, [2] visible, [3] largest depth key

// Radix sort geometry: one workgroup (256 threads = 4 waves) ranks a 2048-element chunk (measured best of 1024/2048/4096 on MI355X);
// each wave owns 512 consecutive elements so that stability needs no cross-wave ordering.
constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS_PER_WAVE = 512;
constexpr int SORT_CHUNK = 4 * SORT_ITEMS_PER_WAVE;
constexpr int SORT_ROUNDS = SORT_ITEMS_PER_WAVE / 64;
constexpr int RADIX_BITS = 8;
constexpr int RADIX_BITS_WIDE = 9;   // the depth sort: 27 significant bits of (key - bits(0.2f)) in three passes
constexpr int RADIX_BINS = 1 << RADIX_BITS;
constexpr uint32_t RADIX_INVALID_KEY = 0xFFFFFFFFu;
// A Gaussian's run of per-instance gradient slots is summed by its owner lane in the backward preprocess up to this length;
// longer runs (screen-filling splats: 3 359 of the 934 k visible Gaussians at C3, the longest 3 192 tiles) are LISTED by the
// forward pass and summed one WAVE per run by a kernel of their own between the two backward kernels (partials.h) --
// densification appends the children of split (large) Gaussians consecutively, and 64 unbounded runs in one wave of
// preprocess_bwd once turned that wave into the kernel's tail (140 -> 450 us after ten densifications at C3).
// (Round 5 also built the alternative without that kernel: the backward blend ADDING such a run's sums into a few shared slots
// with float atomics.  16 us faster at C3, where 1.6 % of the written slots belong to long runs -- and 83 us SLOWER in the
// backward blend of the mapper-loop leg, whose inflated scene has them everywhere: nine atomics per touched instance do not
// scale.  profiles/r05_d, r05_w; reverted.)
constexpr uint32_t LONG_RUN = 64;
constexpr int LONG_LISTS = 64, LONG_COUNT_STRIDE = 32;
// the offset scan's first pass lists them (sort.hip: scan_reduce_rect_kernel; its waves append to the LONG_LISTS sub-lists in
// turn -- one global counter would serialise a few thousand same-address atomics): the capacity of a sub-list covers its appenders
static inline size_t long_list_capacity(size_t P);

// float4 per instance gradient slot: nine sums in three float4 (48 bytes).  A 48-byte slot straddles two 64-byte sectors half of
// the time, and at 22 % density every slot is fetched, and written, on its own; padded to 64 bytes (GSR_EXTRA_FLAGS=-DGSR_SLOT_F4=4)
// the step was 16 us SLOWER on one box (1.506 -> 1.522 ms median, profiles/r05_v: the per-Gaussian stage +7 us, every stage that
// touches the 33 % larger binning buffer a little) -- measured in round 5, not kept.
#ifndef GSR_SLOT_F4
#define GSR_SLOT_F4 3
#endif
constexpr int SLOT_F4 = GSR_SLOT_F4;

constexpr int SCAN_THREADS = 256;

static inline int sort_blocks(int n) { return n > 0 ? div_up(n, SORT_CHUNK) : 1; }
// items per scan block: at least 2048, and large enough that the block-sum spine fits one block pass
static inline int scan_items_per_block(int n)
{
	int ipb = 2048;
	while ((long long)ipb * 1024 < (long long)n) ipb *= 2;
	return ipb;
}
static inline int scan_blocks(int n) { return n > 0 ? div_up(n, scan_items_per_block(n)) : 1; }
static inline size_t long_list_capacity(size_t P)
{
	// every (scan block, trip, wave) appends at most 64 entries to ONE sub-list, and the sub-lists take them in turn
	const size_t ipb = (size_t)scan_items_per_block((int)P), blocks = (P + ipb - 1) / ipb;
	const size_t appenders = blocks * (ipb / SCAN_THREADS) * (SCAN_THREADS / 64);
	return ((appenders + LONG_LISTS - 1) / LONG_LISTS) * 64;
}
static inline size_t scan_scratch_elems(int n) { return (size_t)scan_blocks(n) + 64; }
// histogram [RADIX_BINS][blocks] + its scanned copy + scan spine
static inline size_t sort_scratch_elems(int n)
{
	size_t h = (size_t)RADIX_BINS * sort_blocks(n);
	return 2 * h + scan_scratch_elems((int)h) + 64;
}
// ... of a sort with RADIX_BITS_WIDE-bit digits (the [512][blocks] table + the 512 row totals)
static inline size_t sort_scratch_elems_wide(int n) { return ((size_t)(1 << RADIX_BITS_WIDE) * sort_blocks(n)) + (1 << RADIX_BITS_WIDE) + 64; }
// ... of the tile sort of n instances: the 8-bit passes' tables, or -- a list short enough for the one-pass form on some tile grid
// -- the [2048][blocks] table + 2 048 row totals
static inline size_t tile_sort_scratch_elems(int n)
{
	const size_t a = sort_scratch_elems(n);
	const size_t b = n <= 512 * 1024 ? ((size_t)(1 << 11) * sort_blocks(n)) + (1 << 11) + 64 : 0;
	return a > b ? a : b;
}

// The 48-byte per-Gaussian record the blend kernels gather (3 x float4):
//   q0 = (mean2D.x, mean2D.y, conic.x, conic.y)   q1 = (conic.z, opacity, r, g)
//   q2 = (b, bits(rect.min), bits(rect.max), bits(first instance slot))   [rect packed x | y<<16]
// One record instead of the reference's three separate gathers (means2D / conic_opacity /
// rgb, forward.cu:317-320,355).
constexpr int REC_FLOAT4S = 3;

struct GeometryState {
	uint4*    wave_counts;    // [wave_count_slots(P)] per wave of preprocess_fwd: (tiles touched, visible Gaussians, largest depth key of a
	                          // visible Gaussian, -); the totals are
	                          // num_rendered and gsr_last_visible_count() (summed inside the depth sort's first two launches)
	uint4*    count_partials; // [sort_blocks(P)] the first level of that sum
	uint32_t* long_runs;      // [LONG_LISTS * long_list_capacity(P)] ids of the Gaussians that touch more than LONG_RUN tiles
	uint32_t* long_counts;    // [LONG_LISTS * LONG_COUNT_STRIDE] entries per sub-list, one cache line apart (zeroed by the projection kernel)
	uint32_t  long_capacity;  // long_list_capacity(P)
	uint32_t* depth_key;      // [P]
	uint32_t* tiles_touched;  // [P]
	int*      radii;          // [P]
	uint16_t* rect;           // [4P]
	float4*   rec;            // [3P]
	float*    cov3D;          // [6P]
	uint8_t*  clamped;        // [P]
	uint32_t* order;          // [P]  final depth-sorted ids
	uint32_t* offsets;        // [P]  exclusive scan of tiles_touched in depth order
	uint32_t* sort_keys_a;    // [P]
	uint32_t* sort_keys_b;    // [P]
	uint32_t* sort_vals_b;    // [P]
	uint32_t* sort_scratch;   // [max(sort_scratch_elems(P), sort_scratch_elems_wide(P))]
	uint32_t* scan_scratch;   // [scan_scratch_elems(P)]
	uint2*    rect_sorted;    // [P] the tile rectangles in depth order (entry i belongs to order[i]): gathered once by the offset
	                          // scan, read linearly by the instance emission
	uint32_t* visible;        // [32] [0] number of visible Gaussians V, left by the first pass of the depth sort

	static GeometryState carve(char* chunk, size_t P, size_t* bytes = nullptr)
	{
		GeometryState g;
		Carver c(chunk);
		g.depth_key = c.take<uint32_t>(P);
		g.tiles_touched = c.take<uint32_t>(P);
		g.radii = c.take<int>(P);
		g.rect = c.take<uint16_t>(4 * P);
		g.rec = c.take<float4>(REC_FLOAT4S * P);
		g.cov3D = c.take<float>(6 * P);
		g.clamped = c.take<uint8_t>(P);
		g.order = c.take<uint32_t>(P);
		g.offsets = c.take<uint32_t>(P);
		g.sort_keys_a = c.take<uint32_t>(P);
		g.sort_keys_b = c.take<uint32_t>(P);
		g.sort_vals_b = c.take<uint32_t>(P);
		g.sort_scratch = c.take<uint32_t>(sort_scratch_elems((int)P) > sort_scratch_elems_wide((int)P) ? sort_scratch_elems((int)P) : sort_scratch_elems_wide((int)P));
		g.scan_scratch = c.take<uint32_t>(scan_scratch_elems((int)P));
		g.rect_sorted = c.take<uint2>(P);
		g.visible = c.take<uint32_t>(32);
		g.wave_counts = c.take<uint4>(wave_count_slots(P));   // (written in full by every forward pass: nothing has to be zeroed)
		g.count_partials = c.take<uint4>((size_t)sort_blocks((int)P));
		g.long_runs = c.take<uint32_t>((size_t)LONG_LISTS * long_list_capacity(P));
		g.long_counts = c.take<uint32_t>((size_t)LONG_LISTS * LONG_COUNT_STRIDE);
		g.long_capacity = (uint32_t)long_list_capacity(P);
		if (bytes) *bytes = c.used(chunk) + 128;
		return g;
	}
};

static inline size_t touched_clear_bytes(size_t R) { return (R + 64 + 255) & ~(size_t)255; }
struct BinningState {
	uint32_t* keys_a;        // [R] tile id per instance (ping)
	uint32_t* vals_a;        // [R] Gaussian id per instance (ping)
	uint32_t* keys_b;        // [R] (pong)
	uint32_t* vals_b;        // [R] (pong)
	uint32_t* sort_scratch;  // [tile_sort_scratch_elems(R)]
	float*    partials;      // [4 SLOT_F4 R] per-instance gradient slots of the backward blend (blend.h), indexed by emission order
	uint8_t*  touched;       // [R] 1 where the backward blend wrote the slot (cleared per backward; the slots themselves are not)
	uint8_t*  contrib;       // [4][R] per quad of the tile: 1 where the forward blend found a pixel of the quad that blends the
	                         // list entry (written by blend_fwd for the batches it walks, read by blend_bwd: blend.h)

	static BinningState carve(char* chunk, size_t R, size_t* bytes = nullptr)
	{
		BinningState b;
		Carver c(chunk);
		b.keys_a = c.take<uint32_t>(R);
		b.vals_a = c.take<uint32_t>(R);
		b.keys_b = c.take<uint32_t>(R);
		b.vals_b = c.take<uint32_t>(R);
		b.sort_scratch = c.take<uint32_t>(tile_sort_scratch_elems((int)R));
		b.partials = c.take<float>(4 * (size_t)SLOT_F4 * R);
		// readers fetch flags 16 bytes at a time (+ 64); the clear covers touched_clear_bytes(R): a multiple of 256 bytes, because
		// the runtime splits a memset of any other size into two kernels (body + tail, 5 us each)
		b.touched = c.take<uint8_t>(touched_clear_bytes(R));
		b.contrib = c.take<uint8_t>(4 * R + 64);
		if (bytes) *bytes = c.used(chunk) + 128;
		return b;
	}
};

// The blend kernels' deal of tiles to the XCDs (blend.h)
struct TileDeal {
	int tiles, grid_x;
	int mode;   // > 0: row-major chunks of `mode` tiles, round robin over the XCDs; 0: one band per XCD
};
static inline TileDeal make_tile_deal(int tiles, int grid_x, int mode)
{
	TileDeal d;
	d.tiles = tiles; d.grid_x = grid_x; d.mode = mode < 0 ? 8 : mode;
	return d;
}

struct ImageState {
	float*    final_T;    // [N]
	uint32_t* n_contrib;  // [N]
	uint2*    ranges;     // [T]

	static ImageState carve(char* chunk, size_t N, size_t T, size_t* bytes = nullptr)
	{
		ImageState im;
		Carver c(chunk);
		im.final_T = c.take<float>(N);
		im.n_contrib = c.take<uint32_t>(N);
		im.ranges = c.take<uint2>(T);
		if (bytes) *bytes = c.used(chunk) + 128;
		return im;
	}
};

// getHigherMsb, cuda_rasterizer/rasterizer_impl.cu:35-50: number of tile-id bits the sort covers.
static inline uint32_t higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1) {
		step /= 2;
		if (n >> msb) msb += step;
		else msb -= step;
	}
	if (n >> msb) msb++;
	return msb;
}

// After `passes` ping-pong radix passes starting in buffer A, where does the result live?
static inline bool result_in_a(int passes) { return (passes % 2) == 0; }
// The tile sort's digit width: 8 bits per pass -- or, for a SMALL list (up to TILE_SORT_ONE_PASS_MAX instances) on a tile grid of
// up to 2 048 tiles, ALL tile bits in ONE pass of up to 11 bits (RADIX_BITS_ONE_PASS): the [digit][block] table, whose column-wise
// accesses sink wide digits on large lists (2 048 scattered lines per workgroup: depth_sort 0.101 -> 0.187 ms at C3, round 3), has
// at most 256 columns there, a pass with its two launches goes, and so does identifyTileRanges -- with the whole tile id as the
// digit the ranges ARE the digit totals' prefix sums (the scatter pass writes them).  A view of 50 k Gaussians at 640 x 480 is
// launch-bound: every launch is ~5 us of a 250 us step.  A pure function of (tiles, R): gsr_forward, gsr_backward and the
// test-suite's view of the buffer agree on where the sorted list lives.
constexpr int RADIX_BITS_ONE_PASS = 11;
constexpr int TILE_SORT_ONE_PASS_MAX = 512 * 1024;
static inline int tile_sort_digit_bits(int tiles, int R)
{
	const int bits = (int)higher_msb((uint32_t)tiles);
	return (bits > RADIX_BITS && bits <= RADIX_BITS_ONE_PASS && R <= TILE_SORT_ONE_PASS_MAX) ? RADIX_BITS_ONE_PASS : RADIX_BITS;
}
static inline int tile_sort_passes(int tiles, int R) { return div_up((int)higher_msb((uint32_t)tiles), tile_sort_digit_bits(tiles, R)); }

// ---- device launchers (one per translation unit) ------------------------------------
// n_dev (nullable, with gather only): a device word; elements from *n_dev on count as zeros (their gather indices are undefined)
// exclusive scan of the tile counts of rect[gather[i]] ((maxx - minx) * (maxy - miny), packed as preprocess_fwd writes them) into
// out, with the gathered rectangles left in rect_sorted; elements from *n_dev on count as zeros
// seeds (nullable): seeds[k] = the element whose instances hold instance slot k * seed_stride, for k < seed_capacity
// long_runs / long_counts / long_capacity (nullable): the elements whose count exceeds LONG_RUN are listed (GeometryState)
int launch_scan_rect_tiles(const uint2* rect, const uint32_t* gather, uint32_t* out, uint2* rect_sorted, int n, uint32_t* scratch,
                           hipStream_t stream, const uint32_t* n_dev, uint32_t* seeds = nullptr, uint32_t seed_stride = 1,
                           uint32_t seed_capacity = 0, uint32_t* long_runs = nullptr, uint32_t* long_counts = nullptr,
                           uint32_t long_capacity = 0);
int launch_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int n, bool inclusive,
                    uint32_t* scratch, hipStream_t stream, const uint32_t* n_dev = nullptr);
// Tile-first binning (sort.hip: compact_scan_kernel; tile_depth_sort.hip): the visible Gaussians compacted in id order -- for the
// r-th of them order[r] = its id, offsets[r] = the exclusive sum of the tile counts, rect_sorted[r] = its rectangle (+ the emission's
// seeds and the list of long runs, as launch_scan_rect_tiles) -- from tiles_touched / rect and the projection kernel's per-wave
// counts (wave_counts, n_pairs = wave_count_slots(n); partials: [scan_blocks(n)] scratch).  The totals (HOST_COUNT_WORDS) are stored
// into the mapped host words host_out by a one-workgroup launch IN FRONT of the compaction; `ready` (a hipEvent_t, nullable) is
// recorded right behind it.  visible_out (nullable, a device word): the number of visible Gaussians.
constexpr int COMPACT_ONE_LEVEL_PAIRS = 2048;   // up to this many per-wave pairs (128 k Gaussians) every block sums the pairs in front of it itself
int launch_compact_visible(const uint32_t* tiles_touched, const uint2* rect, const uint4* wave_counts, int n_pairs, uint4* partials,
                           uint32_t* order, uint32_t* offsets, uint2* rect_sorted, int n, uint32_t* host_out, void* ready, hipStream_t stream,
                           uint32_t* seeds = nullptr, uint32_t seed_stride = 1, uint32_t seed_capacity = 0, uint32_t* long_runs = nullptr,
                           uint32_t* long_counts = nullptr, uint32_t long_capacity = 0, uint32_t* visible_out = nullptr);
// every tile's list [ranges[t].x, ranges[t].y) of point_list (ascending id on entry) sorted by depth_key[id], stable: one workgroup
// per tile.  spare_keys / spare_vals / spare_words: three arrays as long as point_list; tile t uses its own segment of each.
int launch_tile_depth_sort(const uint2* ranges, int tiles, const uint32_t* depth_key, uint32_t* point_list, uint32_t* spare_keys,
                           uint32_t* spare_vals, uint32_t* spare_words, hipStream_t stream);
// A job that rides in the first two launches of a sort: add up n (a, b) pairs -- every histogram workgroup its share into
// partials[block], an extra workgroup of the row-prefix launch the partials -- and store the totals, sum a as 64 bits in
// host_out[0..1], sum b in host_out[2], into MAPPED HOST memory; `ready` (a hipEvent_t, nullable) is recorded right behind the
// second launch.  gsr_forward: the per-wave (tiles touched, visible) pairs of the projection kernel.
struct RadixHostCount {
	const uint4* pairs = nullptr;   // (a, b, c, -): sum a (64-bit), sum b, max c
	int n = 0;
	uint4* partials = nullptr;      // [sort_blocks(elements of the sort)]
	uint32_t* host_out = nullptr;   // device address of mapped host memory (HOST_COUNT_WORDS words)
	void* ready = nullptr;
};
// Stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit, end_bit), 8 bits per pass.
// Pass 0 reads (keys_in, vals_in) -- read-only, vals_in == nullptr means value = index -- and writes the
// pong buffers; later passes alternate ping <- pong <- ping.  *keys_res / *vals_res receive the buffers
// holding the result (pong for an odd number of passes, ping for an even one).
int launch_radix_sort(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_ping, uint32_t* vals_ping,
                      uint32_t* keys_pong, uint32_t* vals_pong, int n, int begin_bit, int end_bit,
                      uint32_t* scratch, hipStream_t stream, uint32_t** keys_res, uint32_t** vals_res,
                      uint32_t* compact_count = nullptr, const RadixHostCount* host_count = nullptr, bool first_hist_ready = false,
                      int digit_bits = RADIX_BITS, uint32_t bias = 0, uint2* ranges_out = nullptr);
// ranges_out (nullable; a ONE-pass sort whose digit is the whole key): ranges_out[k] = [first, end) of key k's run in the output
// for every key that occurs (the others are left alone: the caller zeroed them) -- identifyTileRanges without its launch
// the digit width of the first pass (the key bits are spread evenly over the passes) -- for a producer that counts the first
// histogram itself (first_hist_ready: the [digit][sort_blocks(n)] table at the head of `scratch`, digits of key bits [begin_bit, +w))
static inline int radix_first_pass_bits(int begin_bit, int end_bit, int digit_bits = RADIX_BITS)
{
	const int passes = end_bit > begin_bit ? div_up(end_bit - begin_bit, digit_bits) : 0;
	return passes ? div_up(end_bit - begin_bit, passes) : 0;
}
// compact_count (nullable, a device word): keys equal to RADIX_INVALID_KEY are "no element": the first pass drops them and
// leaves the number of remaining elements in *compact_count; the later passes read it and touch that many elements only.
// The result buffers then hold that many sorted pairs followed by undefined content.

}  // namespace gsr
