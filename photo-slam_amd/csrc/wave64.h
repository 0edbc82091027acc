// wave64.h -- wavefront-level primitives for CDNA4 (wave = 64 lanes, SIMD-32 x 2 cycles).
//
// Every function must be called from wave-uniform control flow (all 64 lanes of the wave
// reach the call).  Cross-lane reductions use DPP row operations (quad_perm, row mirrors,
// row_bcast15/31 -- the gfx9/CDNA forms), which cost one VALU issue per step and no LDS
// traffic; ballots land in an SGPR pair.
#pragma once
#include "rt.h"

namespace gsr {

#ifdef GSR_EMU
// Emulator versions: implemented by rendezvous of the wave's lanes (tests/emu/hip_emu.h).
using ::hipemu::wave_ballot;
using ::hipemu::wave_fence;
using ::hipemu::wave_incl_scan_u32;
using ::hipemu::wave_max_u32;
using ::hipemu::wave_reduce9_f32;
using ::hipemu::wave_shfl_u32;
using ::hipemu::wave_sum_u32;
using ::hipemu::wave_uniform_u64;
using ::hipemu::wave_uniform_u32;
using ::hipemu::wave_readlane_f32;
using ::hipemu::wave_readlane_u32;
using ::hipemu::wave_writelane_f32;
using ::hipemu::wave_reduce9_swap_f32;
using ::hipemu::wave_swap9_component;
using ::hipemu::mask_select_f32;
using ::hipemu::mask_select_u32;
using ::hipemu::mask_select0_f32;
#define GSR_OPAQUE_F32(x) asm volatile("" : "+x"(x))
static inline float4 load_stream_f4(const float4* p) { return *p; }
static inline void store_stream_f4(float4* p, const float4 v) { *p = v; }
#define GSR_SCHED_BARRIER() ((void)0)
#else

// Broadcast lane `lane`'s value to the whole wave through an SGPR (v_readlane_b32): the value
// becomes a scalar operand of the following VALU instructions -- no LDS, no VGPR.
__device__ __forceinline__ float wave_readlane_f32(float v, int lane)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ uint32_t wave_readlane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
// Put a wave-uniform value into lane `lane` of a VGPR, other lanes keep `old` (v_cmp once per
// target lane + one v_cndmask per value; v_writelane_b32 would need both operands on the
// constant bus, which gfx9 does not allow).
__device__ __forceinline__ float wave_writelane_f32(float old, float uniform_value, int lane)
{
	return ((int)(threadIdx.x & 63u) == lane) ? uniform_value : old;
}

// Tell the compiler a value is wave-uniform (moves it to SGPRs: scalar loops, scalar LDS addresses).
__device__ __forceinline__ uint32_t wave_uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long wave_uniform_u64(unsigned long long v)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
	return ((unsigned long long)hi << 32) | lo;
}

// the builtin takes the i1 directly: a predicate that already lives in an SGPR pair costs no VALU
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// Selects driven by an explicit 64-bit lane mask held in an SGPR pair (v_cndmask_b32 with a scalar mask operand):
// predicates that are combined with scalar logic (and, andn2, or on ballots) never round-trip through the VALU.
__device__ __forceinline__ float mask_select_f32(unsigned long long mask, float if_set, float if_clear)
{
	float r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
	return r;
}
__device__ __forceinline__ uint32_t mask_select_u32(unsigned long long mask, uint32_t if_set, uint32_t if_clear)
{
	uint32_t r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
	return r;
}
__device__ __forceinline__ float mask_select0_f32(unsigned long long mask, float if_set)
{
	float r;
	asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(if_set), "s"(mask));
	return r;
}

// 16-byte load of data that is read exactly once (streaming: no reuse to keep in the caches).
__device__ __forceinline__ float4 load_stream_f4(const float4* p)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
	return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ void store_stream_f4(float4* p, const float4 v)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	v4f t = {v.x, v.y, v.z, v.w};
	__builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// Make a value opaque to the optimiser (no instruction): stops loop-invariant hoisting of everything computed from it.
#define GSR_OPAQUE_F32(x) asm volatile("" : "+v"(x))
// No memory access may be moved across this point by the optimiser and no instruction by the machine scheduler.
#define GSR_SCHED_BARRIER()                \
	do {                                   \
		asm volatile("" ::: "memory");     \
		__builtin_amdgcn_sched_barrier(0); \
	} while (0)

// Scheduling fence for intra-wave communication through LDS that relies on lock-step
// execution (lanes read, then a leader lane writes).  The hardware issues a wave's LDS
// operations in order, so no instruction is needed -- only the compiler must not move
// memory operations across it.
__device__ __forceinline__ void wave_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_shfl_u32(uint32_t v, int src_lane) { return (uint32_t)__shfl((int)v, src_lane, 64); }

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = true>
__device__ __forceinline__ float dpp_f32(float old_v, float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_v), __builtin_bit_cast(int, v),
	                                                              CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = true>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old_v, uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)old_v, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}

// DPP control encodings (LLVM AMDGPU: DppCtrl)
constexpr int DPP_QUAD_PERM_1032 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_PERM_2301 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_SHR2 = 0x112;
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

// Full-wave float sum; the total is valid in lane 63 (other lanes hold partials).
__device__ __forceinline__ float wave_sum_f32_lane63(float v)
{
	v += dpp_f32<DPP_QUAD_PERM_1032>(0.f, v);
	v += dpp_f32<DPP_QUAD_PERM_2301>(0.f, v);
	v += dpp_f32<DPP_ROW_HALF_MIRROR>(0.f, v);
	v += dpp_f32<DPP_ROW_MIRROR>(0.f, v);
	// every lane of a 16-lane row now holds the row sum
	v += dpp_f32<DPP_ROW_BCAST15, 0xa>(0.f, v);   // rows 1,3 += row 0,2 (lane 15 of previous row)
	v += dpp_f32<DPP_ROW_BCAST31, 0xc>(0.f, v);   // rows 2,3 += lane 31
	return v;
}

// Nine independent full-wave sums at once (the 9 per-Gaussian gradient components of the
// backward blend); results valid in lane 63.  Interleaving the chains hides DPP latency.
__device__ __forceinline__ void wave_reduce9_f32(float (&v)[9])
{
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_QUAD_PERM_1032>(0.f, v[i]);
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_QUAD_PERM_2301>(0.f, v[i]);
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_ROW_HALF_MIRROR>(0.f, v[i]);
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_ROW_MIRROR>(0.f, v[i]);
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_ROW_BCAST15, 0xa>(0.f, v[i]);
#pragma unroll
	for (int i = 0; i < 9; i++) v[i] += dpp_f32<DPP_ROW_BCAST31, 0xc>(0.f, v[i]);
}

// Nine full-wave sums, packed from the TOP of the butterfly.  gfx950's v_permlane32_swap / v_permlane16_swap
// exchange half-waves / odd-even rows BETWEEN two registers, so one swap + one add folds two values
// into one register (value a in the lower half / even rows, value b in the upper half / odd rows):
//   level 32: 8 values -> 4 registers (4 swaps + 2 packed adds), level 16: 4 -> 2 (2 swaps + 1 packed add),
//   level 8 : 2 -> 1 with a DPP row_ror:8 whose bank_mask keeps one value per half row (3 ops),
//   levels 4, 2, 1: half-mirror and quad_perm adds on ONE register (3 ops);
// the ninth value is only summed inside each row of 16 (4 DPP adds; the caller merges the four row sums, e.g.
// with the LDS atomic it issues anyway): 19 VALU for nine sums (72 unpacked; 38 for a butterfly packed from the
// bottom with quad_perm selects, an earlier version).  Result: every lane of the 8-lane group g = lane >> 3 holds in `packed` the total of
// value bitrev3(g) = {0,4,2,6,1,5,3,7}[g]; `ninth_row` holds the sum of v[8] over the lane's row of 16.
__device__ __forceinline__ void wave_reduce9_swap_f32(const float (&v)[9], float& packed, float& ninth_row)
{
	constexpr int DPP_ROW_ROR8 = 0x128;
	typedef float v2f __attribute__((vector_size(8)));
	// The adds of two swaps ride in one v_pk_add_f32, and the operands are paired so that no move is needed between the
	// levels: (p01, p45) = (a0, c0) + (a1, c1), (p23, p67) likewise, then the 16-lane swaps act on (p01, p23) and (p45, p67)
	// in place and one packed add finishes (q0, q1): 9 instructions instead of 12.  Every swap result passes an empty asm:
	// building the vectors straight from the builtins' results makes this compiler drop two of the four swaps.
	float a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3], c0 = v[4], c1 = v[5], d0 = v[6], d1 = v[7];
#define GSR_SWAP(builtin, x, y)                                                                                          \
	do {                                                                                                                 \
		const auto r_ = builtin(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);         \
		x = __builtin_bit_cast(float, (unsigned)r_[0]);                                                                  \
		y = __builtin_bit_cast(float, (unsigned)r_[1]);                                                                  \
		GSR_OPAQUE_F32(x);                                                                                               \
		GSR_OPAQUE_F32(y);                                                                                               \
	} while (0)
	GSR_SWAP(__builtin_amdgcn_permlane32_swap, a0, a1);
	GSR_SWAP(__builtin_amdgcn_permlane32_swap, c0, c1);
	GSR_SWAP(__builtin_amdgcn_permlane32_swap, b0, b1);
	GSR_SWAP(__builtin_amdgcn_permlane32_swap, d0, d1);
	const v2f pa = (v2f){a0, c0} + (v2f){a1, c1};   // (p01, p45): rows 0,1: v0 (row r + row r+2), rows 2,3: v1
	const v2f pb = (v2f){b0, d0} + (v2f){b1, d1};   // (p23, p67)
	float p01 = pa[0], p45 = pa[1], p23 = pb[0], p67 = pb[1];
	GSR_SWAP(__builtin_amdgcn_permlane16_swap, p01, p23);   // rows: v0 v2 v1 v3 (each lane: its column over all four rows)
	GSR_SWAP(__builtin_amdgcn_permlane16_swap, p45, p67);   // rows: v4 v6 v5 v7
#undef GSR_SWAP
	const v2f qq = (v2f){p01, p45} + (v2f){p23, p67};
	const float q0 = qq[0], q1 = qq[1];
	float r = q0 + dpp_f32<DPP_ROW_ROR8>(0.f, q0);
	// lanes 8..15 of every row <- q1 + q1 rotated by 8 (one v_add_f32_dpp whose bank_mask leaves lanes 0..7 alone;
	// the builtin form costs five instructions).  s_nop: a DPP source needs two wait states after its VALU write.
	asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(r) : "v"(q1));
	r += dpp_f32<DPP_ROW_HALF_MIRROR>(0.f, r);
	r += dpp_f32<DPP_QUAD_PERM_2301>(0.f, r);
	r += dpp_f32<DPP_QUAD_PERM_1032>(0.f, r);
	packed = r;
	float n = v[8];
	n += dpp_f32<DPP_QUAD_PERM_1032>(0.f, n);
	n += dpp_f32<DPP_QUAD_PERM_2301>(0.f, n);
	n += dpp_f32<DPP_ROW_HALF_MIRROR>(0.f, n);
	n += dpp_f32<DPP_ROW_MIRROR>(0.f, n);
	ninth_row = n;
}
// the value whose total `packed` holds in this lane
__device__ __forceinline__ int wave_swap9_component(int lane)
{
	const int g = lane >> 3;
	return ((g & 1) << 2) | (g & 2) | ((g >> 2) & 1);
}
// Inclusive prefix sum across the wave (Hillis-Steele on DPP row shifts + row broadcasts).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
	v += dpp_u32<DPP_ROW_SHR1>(0u, v);
	v += dpp_u32<DPP_ROW_SHR2>(0u, v);
	v += dpp_u32<DPP_ROW_SHR4>(0u, v);
	v += dpp_u32<DPP_ROW_SHR8>(0u, v);
	v += dpp_u32<DPP_ROW_BCAST15, 0xa>(0u, v);
	v += dpp_u32<DPP_ROW_BCAST31, 0xc>(0u, v);
	return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
	v = wave_incl_scan_u32(v);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
		v = v > o ? v : o;
	}
	return v;
}
#endif  // GSR_EMU

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ unsigned long long lanemask_lt()
{
	return (1ull << (threadIdx.x & 63u)) - 1ull;
}

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

// For every lane, the set of lanes (among `valid` ones) holding the same `nbits`-bit digit.
__device__ __forceinline__ unsigned long long wave_match_digit(uint32_t digit, int nbits, bool valid)
{
	unsigned long long m = wave_ballot(valid);
	for (int b = 0; b < nbits; b++) {
		const bool bit = (digit >> b) & 1u;
		const unsigned long long s = wave_ballot(valid && bit);
		m &= bit ? s : ~s;
	}
	return m;
}

}  // namespace gsr
