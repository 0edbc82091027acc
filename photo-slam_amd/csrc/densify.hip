// densify.hip -- GaussianModel::densifyAndPrune / prunePoints as stream compaction.
//
// Reference: src/gaussian_model.cpp:588-815.  densifyAndClone appends the clones (cat), densifyAndSplit appends 2 children
// per selected Gaussian (cat) and prunes the parents (boolean-mask index of all six tensors + both Adam moments), the
// final prune masks everything again -- every tensor and moment is copied 4-6 times, each step syncs on a mask count,
// and emptyCache() drops the allocator blocks.  The RESULT is a pure function of per-Gaussian decisions:
//
//   new arrays = [ originals that are neither split nor pruned | clones that survive the prune |
//                  first children that survive | second children that survive ],   every block in source order
//
// (clones carry their source's opacity and scale, so they are pruned iff the source is; both children of a parent share
// its opacity and its scale / 1.6, so they survive or go together; max_radii2D is zeroed by densificationPostfix BEFORE the
// final prune, so the screen-size term of :808 never fires).  Hence:
//
//   gsr_densify_select : one pass over (accum, denom, scaling, opacity) -> five flags per Gaussian, block totals, a spine
//                        scan, and a second pass that ranks every flag with wave ballots (deterministic: no atomics) and
//                        writes the GATHER PLAN: for every row of the new arrays its source row and kind, for every
//                        surviving child its row in the [2k,3] normal-sample matrix.  Counts stay on the device; the host
//                        reads them once (it has to size the new tensors).
//   gsr_densify_gather : one launch rebuilds all five parameter tensors and their ten Adam moment tensors from the plan
//                        (rows copied once; new rows get zero moments), computes the children's positions
//                        R(q) * (z * exp(s)) + xyz and scales log(exp(s) / 1.6) in place of the copy, and zeroes the three
//                        statistics arrays.  708 B read + 708 B written per surviving Gaussian: HBM-bound.
//
// prunePoints(mask) is the same plan with keep = !mask and nothing else.
#include "kernels.h"
#include "wave64.h"

namespace gsr {

constexpr int DS_BLOCK = 256;
// flag bits
constexpr uint32_t F_KEEP = 1u, F_CLONE_KEPT = 2u, F_CHILD_KEPT = 4u, F_SPLIT = 8u, F_CLONE_SEL = 16u;

struct DensifySelect {
	int P;
	const float* accum;
	const float* denom;
	const float* scaling;        // [P,3] log-scales
	const float* opacity;        // [P] logits
	const uint8_t* prune_mask;   // non-null: prunePoints(mask) -- keep = !mask, nothing else
	float max_grad, min_opacity, thr_dense, thr_world;
	int use_world;               // max_screen_size != 0: also prune scale > 0.1 extent (:809)
};

// scale / (0.8 * N) of densifyAndSplit (:738) as ATen evaluates `tensor / 1.6` on the device (multiplication by the fp32
// reciprocal, BinaryDivTrueKernel.cu) resp. on the host (a true division)
__device__ __forceinline__ float child_scale(float s)
{
#ifdef GSR_EMU
	return s / 1.6f;
#else
	return s * (1.0f / 1.6f);
#endif
}

__device__ __forceinline__ uint32_t densify_flags(const DensifySelect& p, int i)
{
	if (p.prune_mask) return p.prune_mask[i] ? 0u : F_KEEP;
	// grads = accum / denom, nan -> 0 (:800-802)
	float g = p.accum[i] / p.denom[i];
	if (g != g) g = 0.f;
	const float s0 = expf(p.scaling[3 * (size_t)i + 0]), s1 = expf(p.scaling[3 * (size_t)i + 1]), s2 = expf(p.scaling[3 * (size_t)i + 2]);
	const float smax = fmaxf(fmaxf(s0, s1), s2);
	const bool big = smax > p.thr_dense;
	// densifyAndClone (:768-773): frobenius_norm over the last dimension of [P,1], i.e. |g| (ATen's norm kernel scales by the
	// maximum: no overflow / underflow of g*g for |g| outside 1e-19 .. 1.8e19); densifyAndSplit (:723-730): g itself
	const bool clone_sel = (fabsf(g) >= p.max_grad) && !big;
	const bool split_sel = (g >= p.max_grad) && big;
	// the final prune (:805-813) on the would-be rows
	const float op = 1.0f / (1.0f + expf(-p.opacity[i]));
	bool prune_self = op < p.min_opacity;
	bool prune_child = prune_self;
	if (p.use_world) {
		prune_self = prune_self || (smax > p.thr_world);
		// the children's activated scale is exp(log(s / 1.6)) (:738 stores the log, :809 activates it again)
		const float c = fmaxf(fmaxf(expf(logf(child_scale(s0))), expf(logf(child_scale(s1)))), expf(logf(child_scale(s2))));
		prune_child = prune_child || (c > p.thr_world);
	}
	uint32_t f = 0;
	if (!split_sel && !prune_self) f |= F_KEEP;
	if (clone_sel) f |= F_CLONE_SEL;
	if (clone_sel && !prune_self) f |= F_CLONE_KEPT;
	if (split_sel) f |= F_SPLIT;
	if (split_sel && !prune_child) f |= F_CHILD_KEPT;
	return f;
}

// pass 1: totals of the five flags per block of 256 Gaussians -> block_counts[5][nblocks]
__global__ void __launch_bounds__(DS_BLOCK)
densify_count_kernel(const DensifySelect p, uint32_t* __restrict__ block_counts, int nblocks)
{
	__shared__ uint32_t s_cnt[DS_BLOCK / 64][5];
	const int i = (int)blockIdx.x * DS_BLOCK + (int)threadIdx.x;
	const uint32_t f = i < p.P ? densify_flags(p, i) : 0u;
	const int w = wave_id();
#pragma unroll
	for (int b = 0; b < 5; b++) {
		const unsigned long long m = wave_ballot((f >> b) & 1u);
		if (lane_id() == 0) s_cnt[w][b] = (uint32_t)__popcll(m);
	}
	__syncthreads();
	if (threadIdx.x < 5) {
		uint32_t t = 0;
		for (int k = 0; k < DS_BLOCK / 64; k++) t += s_cnt[k][threadIdx.x];
		block_counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = t;
	}
}

// pass 2: exclusive scan of every counter over the blocks (one workgroup; nblocks = P / 256), totals -> counts[0..4],
// counts[5] = rows of the new arrays
__global__ void __launch_bounds__(1024)
densify_spine_kernel(uint32_t* __restrict__ block_counts, int nblocks, int* __restrict__ counts)
{
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_carry;
	__shared__ uint32_t s_tot[5];
	for (int b = 0; b < 5; b++) {
		if (threadIdx.x == 0) s_carry = 0;
		__syncthreads();
		uint32_t* row = block_counts + (size_t)b * nblocks;
		for (int base = 0; base < nblocks; base += 1024) {
			const int i = base + (int)threadIdx.x;
			const uint32_t v = i < nblocks ? row[i] : 0u;
			const uint32_t incl = wave_incl_scan_u32(v);
			if (lane_id() == 63) s_wave[wave_id()] = incl;
			__syncthreads();
			uint32_t off = s_carry;
			for (int k = 0; k < wave_id(); k++) off += s_wave[k];
			if (i < nblocks) row[i] = off + incl - v;
			__syncthreads();
			if (threadIdx.x == 1023) s_carry = off + incl;
			__syncthreads();
		}
		if (threadIdx.x == 0) s_tot[b] = s_carry;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		for (int b = 0; b < 5; b++) counts[b] = (int)s_tot[b];
		counts[5] = (int)(s_tot[0] + s_tot[1] + 2u * s_tot[2]);
		counts[6] = 0;
		counts[7] = 0;
	}
}

// pass 3: ranks -> the gather plan.  src_of[row] = source | kind << 30 (0 kept original, 1 clone, 2 first child, 3 second
// child); child_sample[c] = row of child c (counted from the first child row) in the [2k,3] sample matrix: the reference
// draws samples for ALL selected parents, k = counts[3], parent-major within each copy (repeat({N,1}), :732-735).
__global__ void __launch_bounds__(DS_BLOCK)
densify_plan_kernel(const DensifySelect p, const uint32_t* __restrict__ block_offs, int nblocks, const int* __restrict__ counts,
                    uint32_t* __restrict__ src_of, uint32_t* __restrict__ child_sample)
{
	__shared__ uint32_t s_cnt[DS_BLOCK / 64][4];
	const int i = (int)blockIdx.x * DS_BLOCK + (int)threadIdx.x;
	const uint32_t f = i < p.P ? densify_flags(p, i) : 0u;
	const int w = wave_id();
	uint32_t rank[4];
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const unsigned long long m = wave_ballot((f >> b) & 1u);
		rank[b] = (uint32_t)__popcll(m & lanemask_lt());
		if (lane_id() == 0) s_cnt[w][b] = (uint32_t)__popcll(m);
	}
	__syncthreads();
#pragma unroll
	for (int b = 0; b < 4; b++) {
		uint32_t off = block_offs[(size_t)b * nblocks + blockIdx.x];
		for (int k = 0; k < w; k++) off += s_cnt[k][b];
		rank[b] += off;
	}
	if (i >= p.P) return;
	const uint32_t n_keep = (uint32_t)counts[0], n_clone = (uint32_t)counts[1], n_child = (uint32_t)counts[2], k_split = (uint32_t)counts[3];
	if (f & F_KEEP) src_of[rank[0]] = (uint32_t)i;
	if (f & F_CLONE_KEPT) src_of[n_keep + rank[1]] = (uint32_t)i | (1u << 30);
	if (f & F_CHILD_KEPT) {
		const uint32_t c = rank[2];
		src_of[n_keep + n_clone + c] = (uint32_t)i | (2u << 30);
		src_of[n_keep + n_clone + n_child + c] = (uint32_t)i | (3u << 30);
		child_sample[c] = rank[3];
		child_sample[n_child + c] = k_split + rank[3];
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// gather: one launch, work item = one 16-byte (or 4-byte) unit of one destination row of one tensor.  The five tensors are
// laid out as consecutive ranges of work items; each item copies parameter, exp_avg and exp_avg_sq.
struct GatherTensor {
	const float* src[3];   // parameter, exp_avg, exp_avg_sq (moments may be null: no optimizer state)
	float* dst[3];
	int row_floats;        // 3, 48, 1, 3, 4
};
struct DensifyGather {
	int n_new, n_first_child;     // rows; first row of the children block = n_keep + n_clone_kept
	const uint32_t* src_of;
	const uint32_t* child_sample;
	const uint32_t* row_perm;     // null, or: row r of the new set is row row_perm[r] of the plan (gsr_densify_gather_args.morton_scratch)
	const float* samples;         // [2k,3] standard normal draws (at::normal's randn before the scale), null if no children
	GatherTensor t[5];            // xyz, features, opacity, scaling, rotation
	float* stats[3];              // xyz_gradient_accum, denom, max_radii2D of the new set: zero-filled (may be null)
	const int* exist_in;          // exist_since_iter_ [P] (nullable, with exist_out): every new row inherits its source's value
	int* exist_out;               //   (prunePoints :636, densifyAndSplit :744, densifyAndClone :782)
	long long items_before[7];    // prefix sums of the work items: t[0..4], then the statistics (+ exist_since_iter), then the end
};

__device__ __forceinline__ void quat_rotate(const float* q, float sx, float sy, float sz, float& ox, float& oy, float& oz)
{
	// general_utils::build_rotation (include/general_utils.h:33-57) + bmm(R, s)
	const float norm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
	const float r = q[0] / norm, x = q[1] / norm, y = q[2] / norm, z = q[3] / norm;
	const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
	const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
	const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
	ox = R00 * sx + R01 * sy + R02 * sz;
	oy = R10 * sx + R11 * sy + R12 * sz;
	oz = R20 * sx + R21 * sy + R22 * sz;
}

__global__ void __launch_bounds__(256)
densify_gather_kernel(const DensifyGather p)
{
	const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
	if (item >= p.items_before[6]) return;
	if (item >= p.items_before[5]) {   // statistics of the new set start from zero (densificationPostfix :709-711)
		const long long e = item - p.items_before[5];
		const int a = (int)(e / p.n_new);
		const long long row = e - (long long)a * p.n_new;
		if (a == 3) {
			if (p.exist_out) p.exist_out[row] = p.exist_in[p.src_of[p.row_perm ? p.row_perm[row] : (uint32_t)row] & 0x3FFFFFFFu];
		} else if (p.stats[a]) {
			p.stats[a][row] = 0.f;
		}
		return;
	}
	int ti = 0;
#pragma unroll
	for (int k = 1; k < 5; k++) ti += item >= p.items_before[k];
	const GatherTensor& t = p.t[ti];
	const long long local = item - p.items_before[ti];
	const bool vec = (t.row_floats & 3) == 0;            // rows of whole float4s: features (12), rotation (1)
	const int units = vec ? t.row_floats / 4 : t.row_floats;
	const int row = (int)(local / units), u = (int)(local - (long long)row * units);
	const int prow = p.row_perm ? (int)p.row_perm[row] : row;   // the plan's row (its position decides a child's sample)
	const uint32_t code = p.src_of[prow];
	const uint32_t src = code & 0x3FFFFFFFu, kind = code >> 30;
	if (vec) {
		const size_t so = ((size_t)src * units + u), dn = ((size_t)row * units + u);
		reinterpret_cast<float4*>(t.dst[0])[dn] = reinterpret_cast<const float4*>(t.src[0])[so];
		const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
		if (t.dst[1]) reinterpret_cast<float4*>(t.dst[1])[dn] = kind ? zero : reinterpret_cast<const float4*>(t.src[1])[so];
		if (t.dst[2]) reinterpret_cast<float4*>(t.dst[2])[dn] = kind ? zero : reinterpret_cast<const float4*>(t.src[2])[so];
		return;
	}
	const size_t so = (size_t)src * units + u, dn = (size_t)row * units + u;
	float v = t.src[0][so];
	if (kind >= 2u && (ti == 0 || ti == 3)) {
		// a split child (:731-738): position sampled from the parent, scale / 1.6
		const float* ps = p.t[3].src[0] + 3 * (size_t)src;     // the parent's log-scales
		if (ti == 3) {
			v = logf(child_scale(expf(v)));
		} else {
			const float* z = p.samples + 3 * (size_t)p.child_sample[prow - p.n_first_child];
			// at::normal(mean 0, std): randn * std (+ 0)
			const float sx = z[0] * expf(ps[0]), sy = z[1] * expf(ps[1]), sz = z[2] * expf(ps[2]);
			float o[3];
			quat_rotate(p.t[4].src[0] + 4 * (size_t)src, sx, sy, sz, o[0], o[1], o[2]);
			v = o[u] + v;
		}
	}
	t.dst[0][dn] = v;
	if (t.dst[1]) t.dst[1][dn] = kind ? 0.f : t.src[1][so];
	if (t.dst[2]) t.dst[2][dn] = kind ? 0.f : t.src[2][so];
}

// the position every row of the plan is ordered by: its source's (a child: its parent's)
__global__ void __launch_bounds__(256)
densify_plan_positions_kernel(int n_new, const uint32_t* __restrict__ src_of, const float* __restrict__ xyz, float* __restrict__ pos)
{
	const int r = (int)blockIdx.x * 256 + (int)threadIdx.x;
	if (r >= n_new) return;
	const size_t s = (size_t)(src_of[r] & 0x3FFFFFFFu);
	pos[3 * (size_t)r] = xyz[3 * s];
	pos[3 * (size_t)r + 1] = xyz[3 * s + 1];
	pos[3 * (size_t)r + 2] = xyz[3 * s + 2];
}

static inline int densify_blocks(int P) { return div_up(P, DS_BLOCK); }

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_densify_morton_scratch_bytes(int n_new) { return knn_scratch_bytes(n_new < 0 ? 0 : n_new); }

size_t gsr_densify_scratch_bytes(int P)
{
	if (P < 0) P = 0;
	// block counters [5][nblocks], src_of [2P] (the new set has at most P + P rows), child_sample [2P]
	return align_up((size_t)5 * densify_blocks(P) * sizeof(uint32_t), 128) + 2 * align_up((size_t)2 * P * sizeof(uint32_t), 128) + 256;
}

int gsr_densify_select(const gsr_densify_select_args* a, char* scratch, int* counts, void* stream_)
{
	if (!a || !counts || a->P < 0) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	if (a->P == 0) {
		GSR_HIP(hipMemsetAsync(counts, 0, 8 * sizeof(int), stream));
		return GSR_OK;
	}
	if (!scratch) return GSR_ERR_INVALID_ARG;
	if (!a->prune_mask && (!a->xyz_gradient_accum || !a->denom || !a->scaling || !a->opacity)) return GSR_ERR_INVALID_ARG;
	DensifySelect p;
	p.P = a->P;
	p.accum = a->xyz_gradient_accum; p.denom = a->denom; p.scaling = a->scaling; p.opacity = a->opacity;
	p.prune_mask = a->prune_mask;
	p.max_grad = a->max_grad; p.min_opacity = a->min_opacity;
	p.thr_dense = a->percent_dense * a->extent;   // float products, as src/gaussian_model.cpp:729,772,809
	p.thr_world = 0.1f * a->extent;
	p.use_world = a->max_screen_size != 0;
	const int nb = densify_blocks(a->P);
	Carver c(scratch);
	uint32_t* block_counts = c.take<uint32_t>((size_t)5 * nb);
	uint32_t* src_of = c.take<uint32_t>((size_t)2 * a->P);
	uint32_t* child_sample = c.take<uint32_t>((size_t)2 * a->P);
	GSR_LAUNCH(densify_count_kernel, nb, DS_BLOCK, stream, p, block_counts, nb);
	GSR_LAUNCH(densify_spine_kernel, 1, 1024, stream, block_counts, nb, counts);
	GSR_LAUNCH(densify_plan_kernel, nb, DS_BLOCK, stream, p, (const uint32_t*)block_counts, nb, (const int*)counts, src_of,
	           child_sample);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

int gsr_densify_gather(const gsr_densify_gather_args* a, const char* scratch, void* stream_)
{
	if (!a || a->P < 0 || a->n_new < 0 || a->n_keep < 0 || a->n_clone < 0 || a->n_child < 0 || a->n_split < 0) return GSR_ERR_INVALID_ARG;
	if ((long long)a->n_keep + a->n_clone + 2ll * a->n_child != a->n_new || a->n_new > 2ll * a->P) return GSR_ERR_INVALID_ARG;
	if (a->n_new == 0) return GSR_OK;
	if (!scratch || a->P == 0) return GSR_ERR_INVALID_ARG;
	if (a->n_child && !a->samples) return GSR_ERR_INVALID_ARG;
	static const int row_floats[5] = {3, 0, 1, 3, 4};
	DensifyGather p;
	p.n_new = a->n_new;
	p.n_first_child = a->n_keep + a->n_clone;
	Carver c(const_cast<char*>(scratch));
	const int nb = densify_blocks(a->P);
	(void)c.take<uint32_t>((size_t)5 * nb);
	p.src_of = c.take<uint32_t>((size_t)2 * a->P);
	p.child_sample = c.take<uint32_t>((size_t)2 * a->P);
	p.samples = a->samples;
	long long items = 0;
	for (int i = 0; i < 5; i++) {
		GatherTensor& t = p.t[i];
		t.row_floats = i == 1 ? a->features_row_floats : row_floats[i];
		if (t.row_floats <= 0) return GSR_ERR_INVALID_ARG;
		t.src[0] = a->param_in[i]; t.src[1] = a->exp_avg_in[i]; t.src[2] = a->exp_avg_sq_in[i];
		t.dst[0] = a->param_out[i]; t.dst[1] = a->exp_avg_out[i]; t.dst[2] = a->exp_avg_sq_out[i];
		if (!t.src[0] || !t.dst[0]) return GSR_ERR_INVALID_ARG;
		if ((t.src[1] == nullptr) != (t.dst[1] == nullptr) || (t.src[2] == nullptr) != (t.dst[2] == nullptr)) return GSR_ERR_INVALID_ARG;
		if ((t.row_floats & 3) == 0) {   // float4 rows need 16-byte aligned bases
			uintptr_t m = 0;
			for (int k = 0; k < 3; k++) m |= reinterpret_cast<uintptr_t>(t.src[k]) | reinterpret_cast<uintptr_t>(t.dst[k]);
			if (m & 15) return GSR_ERR_UNSUPPORTED;
		}
		p.items_before[i] = items;
		items += (long long)a->n_new * ((t.row_floats & 3) == 0 ? t.row_floats / 4 : t.row_floats);
	}
	p.items_before[5] = items;
	for (int k = 0; k < 3; k++) p.stats[k] = a->stats_out[k];
	if ((a->exist_since_iter_in == nullptr) != (a->exist_since_iter_out == nullptr)) return GSR_ERR_INVALID_ARG;
	p.exist_in = a->exist_since_iter_in; p.exist_out = a->exist_since_iter_out;
	items += 4ll * a->n_new;
	p.items_before[6] = items;
	const long long blocks = (items + 255) / 256;
	if (blocks > 0x7FFFFFFFll) return GSR_ERR_UNSUPPORTED;
	p.row_perm = nullptr;
	if (a->morton_scratch) {
		// the plan's rows along the Z-order curve of their sources' positions (the simple-knn front half: bounding box, Morton codes,
		// radix sort); the gather below then reads the plan through the permutation
		float* pos = morton_points_buffer(a->n_new, a->morton_scratch);
		GSR_LAUNCH(densify_plan_positions_kernel, div_up(a->n_new, 256), 256, (hipStream_t)stream_, a->n_new, p.src_of, a->param_in[0], pos);
		uint32_t* perm = nullptr;
		const int st = launch_morton_order(a->n_new, pos, &perm, a->morton_scratch, (hipStream_t)stream_);
		if (st != GSR_OK) return st;
		p.row_perm = perm;
	}

	GSR_LAUNCH(densify_gather_kernel, (int)blocks, 256, (hipStream_t)stream_, p);
	GSR_CHECK_LAUNCH();
	return GSR_OK;
}

}  // extern "C"
