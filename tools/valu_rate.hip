// tools/valu_rate.hip -- issue rate of the wave64 VALU instructions the blend kernels are made of, on gfx950.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > profiles/rNN_valu_rate.json
//
// Every kernel runs `iters` times an unrolled block of UNROLL copies of ONE instruction on 8 independent register chains
// (so that neither dependency latency nor the loop overhead bounds it), with `w` waves resident on every SIMD
// (grid = 256 CUs x w workgroups of 256 threads = 4 waves, one per SIMD).  The figure reported is
//     cycles per wave64 instruction per SIMD = kernel time x shader clock / (instructions per wave x waves per SIMD),
// the clock being measured by the same launch through s_memtime (shader-clock counter) against the HIP-event time.
// the guide says a wave64 VALU instruction issues over 2 cycles on CDNA4's 32-wide SIMDs;
// tools/gpu_sq.sh assumed 4.  The numbers decide how SQ_INSTS_VALU is turned into an issue utilisation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64;   // instructions per loop body (8 chains x 8)

enum Op { FMA, PK_FMA, MUL, PK_MUL, ADD, PK_ADD, EXP, RCP, ADD_DPP, MOV_DPP, PERMLANE32_SWAP, PERMLANE16_SWAP, CNDMASK_SGPR, CMP,
          FMA_DEP, MAD_U32_24, CNDMASK_VCC, CMP_VCC, CMP_SGPR4, MIN, SUB, FMAC, MOV_SGPR, LSHL_ADD, MFMA_INDEP, MFMA_DEP, MFMA_PLUS_8FMA,
          DS_READ_B128_BCAST, DS_WRITE_B32, N_OPS };
static const char* const OP_NAME[N_OPS] = {
	"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
	"v_add_f32_dpp_row_shr", "v_mov_b32_dpp_row_shr", "v_permlane32_swap", "v_permlane16_swap", "v_cndmask_b32_sgpr_mask",
	"v_cmp_gt_f32_sgpr_dst", "v_fma_f32_one_dependent_chain", "v_mad_u32_u24", "v_cndmask_b32_e32_vcc", "v_cmp_gt_f32_e32_vcc",
	"v_cmp_gt_f32_e64_4_sgpr_pairs", "v_min_f32", "v_sub_f32", "v_fmac_f32_e32", "v_mov_b32_from_sgpr", "v_lshl_add_u32",
	"v_mfma_f32_16x16x4_f32_4_accumulators", "v_mfma_f32_16x16x4_f32_dependent", "1_mfma_16x16x4_plus_8_v_fma_per_9_slots",
	"ds_read_b128_same_address", "ds_write_b32_lane_consecutive"};

// one instruction of the block on chain register(s) `a` (and `b` for the 64-bit operands of the packed forms)
#define ONE(ASM, a) asm volatile(ASM : "+v"(a) : "v"(k0), "v"(k1))
#define ONE2(ASM, a) asm volatile(ASM : "+v"(a) : "v"(kk0), "v"(kk1))

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float* sink, unsigned long long* clocks, int iters)
{
	float k0 = 1.0000001f + sink[0], k1 = 1e-9f + sink[1];
	typedef float float2v __attribute__((ext_vector_type(2)));
	float2v kk0 = {k0, k0}, kk1 = {k1, k1};
	float r[8];
	float2v p[8];
	for (int i = 0; i < 8; i++) { r[i] = (float)(threadIdx.x + i) * 1e-3f; p[i] = float2v{r[i], r[i] + 1.f}; }
	unsigned long long mask = 0x5555555555555555ull + (unsigned long long)sink[2];
	unsigned long long m4[4] = {0, 0, 0, 0};
	typedef float float4v __attribute__((ext_vector_type(4)));
	float4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
	__shared__ float4v lds[256 + 16];
	lds[threadIdx.x] = float4v{k0, k1, k0, k1};
	__syncthreads();
	const unsigned lds_addr = (unsigned)(threadIdx.x / 64) * 16u * 16u, lds_waddr = (unsigned)threadIdx.x * 4u;
	unsigned sgpr_val = (unsigned)sink[2] + 3u;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < UNROLL / 8; u++) {
#pragma unroll
			for (int c = 0; c < 8; c++) {
				if constexpr (OP == FMA) ONE("v_fma_f32 %0, %0, %1, %2", r[c]);
				else if constexpr (OP == FMA_DEP) ONE("v_fma_f32 %0, %0, %1, %2", r[0]);
				else if constexpr (OP == PK_FMA) ONE2("v_pk_fma_f32 %0, %0, %1, %2", p[c]);
				else if constexpr (OP == MUL) ONE("v_mul_f32 %0, %0, %1", r[c]);
				else if constexpr (OP == PK_MUL) ONE2("v_pk_mul_f32 %0, %0, %1", p[c]);
				else if constexpr (OP == ADD) ONE("v_add_f32 %0, %0, %2", r[c]);
				else if constexpr (OP == PK_ADD) ONE2("v_pk_add_f32 %0, %0, %2", p[c]);
				else if constexpr (OP == EXP) ONE("v_exp_f32 %0, %0", r[c]);
				else if constexpr (OP == RCP) ONE("v_rcp_f32 %0, %0", r[c]);
				else if constexpr (OP == ADD_DPP) ONE("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", r[c]);
				else if constexpr (OP == MOV_DPP) ONE("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", r[c]);
				else if constexpr (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[c]), "+v"(r[(c + 4) & 7]));
				else if constexpr (OP == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[c]), "+v"(r[(c + 4) & 7]));
				else if constexpr (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(k0), "s"(mask));
				else if constexpr (OP == CMP) { unsigned long long m; asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(r[c]), "v"(k0)); mask ^= m; }
				else if constexpr (OP == MAD_U32_24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[c]) : "v"(k0), "v"(k1));
				else if constexpr (OP == CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(k0) : "vcc");
				else if constexpr (OP == CMP_VCC) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(r[c]), "v"(k0) : "vcc");
				else if constexpr (OP == CMP_SGPR4) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m4[c & 3]) : "v"(r[c]), "v"(k0));
				else if constexpr (OP == MIN) ONE("v_min_f32 %0, %0, %1", r[c]);
				else if constexpr (OP == SUB) ONE("v_sub_f32 %0, %0, %2", r[c]);
				else if constexpr (OP == FMAC) ONE("v_fmac_f32_e32 %0, %1, %2", r[c]);
				else if constexpr (OP == MOV_SGPR) asm volatile("v_mov_b32 %0, %1" : "=v"(r[c]) : "s"(sgpr_val));
				else if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[c]) : "v"(k0));
				else if constexpr (OP == MFMA_INDEP) acc[c & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[c], k0, acc[c & 3], 0, 0, 0);
				else if constexpr (OP == MFMA_DEP) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[c], k0, acc[0], 0, 0, 0);
				else if constexpr (OP == MFMA_PLUS_8FMA) {
					if (c == 0 && (u & 0) == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, k0, acc[u & 3], 0, 0, 0);
					ONE("v_fma_f32 %0, %0, %1, %2", r[c]);
				}
				else if constexpr (OP == DS_READ_B128_BCAST) { float4v t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lds_addr)); asm volatile("" :: "v"(t)); }
				else if constexpr (OP == DS_WRITE_B32) asm volatile("ds_write_b32 %0, %1" : : "v"(lds_waddr), "v"(r[c]) : "memory");
			}
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if constexpr (OP == DS_READ_B128_BCAST) asm volatile("s_waitcnt lgkmcnt(0)");
	float accs = 0.f;
	for (int i = 0; i < 8; i++) accs += r[i] + p[i].x + p[i].y;
	for (int i = 0; i < 4; i++) accs += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	if (accs == 123.456f || mask == 42 || (m4[0] ^ m4[1] ^ m4[2] ^ m4[3]) == 77) sink[3] = accs;   // keeps the chains alive
	if ((threadIdx.x & 63) == 0) clocks[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
static void run(float* sink, unsigned long long* clocks, int waves_per_simd, int iters, int n_cu, double& ms, double& wave_clocks)
{
	const int grid = n_cu * waves_per_simd;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	rate_kernel<OP><<<grid, 256>>>(sink, clocks, iters / 8);   // warm-up (code object load, clocks up)
	CHECK(hipEventRecord(e0));
	rate_kernel<OP><<<grid, 256>>>(sink, clocks, iters);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float t;
	CHECK(hipEventElapsedTime(&t, e0, e1));
	ms = t;
	std::vector<unsigned long long> h(grid * 4);
	CHECK(hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost));
	double s = 0;
	for (auto c : h) s += (double)c;
	wave_clocks = s / h.size();
	CHECK(hipEventDestroy(e0));
	CHECK(hipEventDestroy(e1));
}

typedef void (*run_fn)(float*, unsigned long long*, int, int, int, double&, double&);
template <int... I> static void fill(run_fn* t, std::integer_sequence<int, I...>) { ((t[I] = run<I>), ...); }

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	float* sink;
	unsigned long long* clocks;
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMemset(sink, 0, 64));
	CHECK(hipMalloc(&clocks, sizeof(unsigned long long) * n_cu * 8 * 4));
	run_fn table[N_OPS];
	fill(table, std::make_integer_sequence<int, N_OPS>{});
	const int iters = 2048;
	printf("{\n \"device\": \"%s\", \"gcnArch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"unroll\": %d, \"iters\": %d,\n", prop.name,
	       prop.gcnArchName, n_cu, prop.clockRate, UNROLL, iters);
	printf(" \"method\": \"cycles per wave64 instruction per SIMD = wave clocks (s_memtime delta, mean over waves) / (instructions per wave x waves per SIMD); event_cycles = the same from the HIP-event time x clockRate\",\n \"ops\": {\n");
	for (int op = 0; op < N_OPS; op++) {
		printf("  \"%s\": {", OP_NAME[op]);
		const int ws[4] = {1, 2, 4, 8};
		for (int wi = 0; wi < 4; wi++) {
			double ms, wc;
			table[op](sink, clocks, ws[wi], iters, n_cu, ms, wc);
			const double insts = (double)iters * UNROLL;
			const double cyc_memtime = wc / (insts * ws[wi]);
			const double cyc_event = ms * 1e-3 * (double)prop.clockRate * 1e3 / (insts * ws[wi]);
			printf("%s\"w%d\": {\"memtime_cycles\": %.3f, \"event_cycles\": %.3f, \"ms\": %.4f}", wi ? ", " : "", ws[wi], cyc_memtime,
			       cyc_event, ms);
		}
		printf("}%s\n", op + 1 < N_OPS ? "," : "");
	}
	printf(" }\n}\n");
	return 0;
}
