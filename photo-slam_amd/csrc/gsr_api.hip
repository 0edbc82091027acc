// gsr_api.hip -- the C-ABI of libgsr_hip.so (include/gsr.h):
// argument validation, scratch carving, and the launch sequence of the forward and
// backward passes on the caller's HIP stream.
//
// Forward launch sequence (reference: Rasterizer::forward, rasterizer_impl.cu:198-336):
//   preprocess (every wave leaves its (tiles, visible) pair with a plain store)
//   -> depth sort of the P Gaussians (3 x 9-bit passes over key - bits(0.2f); its first two launches sum the pairs on the side = num_rendered, into
//      mapped host memory, event behind them -- the reference blocks on a copy here, :281) -> exclusive scan in depth order
//   -> host waits for the event only now, sizes the binning buffer
//   -> emit instances -> tile-id sort over R (ceil(msb(T)/8) passes) -> tile ranges -> blend.
// Tile-first binning (binning_tile_first below; small and mid-size views): no depth sort of the Gaussians -- the visible ones are
// compacted in id order (the count for the host by a one-workgroup launch in front of the compaction), and behind the tile sort
// every tile's list is sorted by depth on its own (tile_depth_sort.hip): 12 launches instead of 20, the same lists bit for bit.
#include <atomic>
#include <chrono>
#include <cmath>
#include <vector>
#include "kernels.h"
#include "shrows.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace gsr {

static thread_local int t_last_err = 0;
static thread_local char t_last_what[256] = {0};
void set_last_hip_error(int err, const char* what)
{
	t_last_err = err;
	snprintf(t_last_what, sizeof(t_last_what), "%s: %s", what ? what : "?", hipGetErrorString((hipError_t)err));
}

// Pinned words + event for the single device->host read of num_rendered, and the second stream with its events: one set per host
// thread AND device (a thread that renders on a second device -- a viewer on another GPU, a test harness -- must not wait on the
// first device's event or launch on its stream), created on first use on the device that is current then, released when the
// thread exits.
struct HostSync {
	uint32_t* pinned = nullptr;
	uint32_t* pinned_dev = nullptr;   // the same words as the device addresses them
	hipEvent_t ev = nullptr;
	// second stream for HBM-bound work that runs next to the VALU-bound blend (the culled rows of the fused SH Adam step, gsr_backward), with the
	// events that fork it from and join it to the caller's stream
	hipStream_t side = nullptr;
	hipEvent_t fork = nullptr, join = nullptr;
	hipEvent_t notify = nullptr;   // gsr_backward_args.color_view_ready_stream: "dL_dcolor_view is complete"
	int init_notify()
	{
		if (!notify) GSR_HIP(hipEventCreateWithFlags(&notify, hipEventDisableTiming));
		return GSR_OK;
	}
	int init_side()
	{
		if (!side) {
			// lowest priority: the blend kernel on the caller's stream keeps the first claim on wave slots
			int lo = 0, hi = 0;
			GSR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
			GSR_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
		}
		if (!fork) GSR_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
		if (!join) GSR_HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming));
		return GSR_OK;
	}
	int init()
	{
		if (pinned && pinned_dev && ev) return GSR_OK;
		if (!pinned) {
			// mapped: the device stores the forward pass's counts straight into it (sort.hip: RadixHostCount)
			GSR_HIP(hipHostMalloc((void**)&pinned, 64 * sizeof(uint32_t), hipHostMallocMapped));
		}
		// (a failure of any step is retried by the next call: a half-initialised set is never used)
		if (!pinned_dev) GSR_HIP(hipHostGetDevicePointer((void**)&pinned_dev, pinned, 0));
		if (!ev) GSR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		return GSR_OK;
	}
	void release()
	{
		if (ev) (void)hipEventDestroy(ev);
		if (fork) (void)hipEventDestroy(fork);
		if (join) (void)hipEventDestroy(join);
		if (notify) (void)hipEventDestroy(notify);
		if (side) (void)hipStreamDestroy(side);
		if (pinned) (void)hipHostFree(pinned);
		*this = HostSync();
	}
};
constexpr int MAX_DEVICES = 16;
struct ThreadState {
	HostSync dev[MAX_DEVICES];
	~ThreadState()
	{
		int cur = 0;
		if (hipGetDevice(&cur) != hipSuccess) return;   // (no runtime left to talk to)
		for (int d = 0; d < MAX_DEVICES; d++)
			if (dev[d].pinned || dev[d].ev || dev[d].side) {
				if (hipSetDevice(d) == hipSuccess) dev[d].release();
			}
		(void)hipSetDevice(cur);
	}
};
static thread_local ThreadState t_state;
// the calling thread's set for the CURRENT device (the device the caller's stream belongs to: PyTorch and the reference keep it current)
static gsr_status host_sync(HostSync** out)
{
	int d = 0;
	GSR_HIP(hipGetDevice(&d));   // (no ROCm device at all: the library's own HIP error, as every later call would give)
	if (d < 0 || d >= MAX_DEVICES) return GSR_ERR_UNSUPPORTED;
	*out = &t_state.dev[d];
	return GSR_OK;
}
static thread_local int t_last_visible = -1;   // gsr_last_visible_count()
// gsr_host_wait_stats(): how long the calling thread was blocked in gsr_forward's ONE host synchronisation (the instance count)
static thread_local double t_sync_wait_us = 0.0;
static thread_local long long t_sync_waits = 0;
// Scheduling switches live in the caller's struct (gsr_sh_adam: no_side_stream, lazy_slice_late, side_blocks; zero = the
// measured-best arrangement).  The environment variable of a switch, when SET, overrides the field -- the A/B handle of the
// bench sessions; read once per process.
static int env_int(const char* name, int unset)
{
	const char* e = getenv(name);
	if (!(e && *e)) return unset;
	// (every caller keeps the value in a function-local static: this line appears once per switch and process)
	fprintf(stderr, "[gsr] environment override %s=%s (wins over the API field / the library's default)\n", name, e);
	return atoi(e);
}
static bool side_stream_enabled(const gsr_sh_adam* o)
{
	static const int env = env_int("GSR_SH_ADAM_SIDE_STREAM", -1);
	return env >= 0 ? env != 0 : !(o && o->no_side_stream);
}
static bool lazy_slice_early(const gsr_sh_adam* o)
{
	static const int env = env_int("GSR_LAZY_SLICE_EARLY", -1);
	return env >= 0 ? env != 0 : !(o && o->lazy_slice_late);
}
// GSR_COV3D_STORED=1 (A/B handle only): the round-3 arrangement -- the forward pass stores the 3-D covariances, the backward pass
// gathers them -- instead of recomputing them in preprocess_bwd (kernels.h: compute_cov3D); same bits either way
static bool cov3D_stored()
{
	static const int env = env_int("GSR_COV3D_STORED", 0);
	return env != 0;
}
// GSR_EMIT_SEEDS=0 (A/B handle): the instance emission searches the offsets for every window instead of starting from the seeds
// the offset scan left (binning.hip); same instances either way
static bool emit_seeded()
{
	static const int env = env_int("GSR_EMIT_SEEDS", 1);
	return env != 0;
}
// The blend kernels' deal of tiles to the XCDs (blend.h: TileDeal::mode); GSR_XCD_CHUNK overrides: 0 = one band per XCD,
// n > 0 = row-major chunks of n tiles.  Measured at C3, alternating runs on one box each (profiles/r05_h, r05_i):
//   one band per XCD            1.609-1.620 ms per step   blend_fwd 0.191  blend_bwd 0.451 ms   TCC traffic of the two 182 / 313 MB
//   row-major chunks of 128     1.589-1.592
//   row-major chunks of 32      1.582-1.588                                                                            229 / 380
//   row-major chunks of 8       1.566-1.575               0.173            0.430                                        261 / 425
//   chunks of 4 / 2 / 1         as 8
// The balance of the deal decides, not the L2 locality.
static int xcd_deal_mode(int tiles)
{
	static const int env = env_int("GSR_XCD_CHUNK", -1);
	if (env >= 0) return env;
	// (small images: at least ~16 rounds of the deal, or its last round is the imbalance)
	int c = 8;
	while (c > 1 && tiles < 8 * 16 * c) c >>= 1;
	return c;
}
// The depth sort's significant bits when it runs on key - DEPTH_KEY_BIAS with 9-bit digits (27 = three passes); 0 = the plain
// sort of 32 bits in four passes.  GSR_DEPTH_SORT_9BIT=0 selects the plain sort (A/B handle); GSR_DEPTH_SORT_BITS=n (tests)
// narrows the range so that ordinary scenes take the re-sort path.
static int depth_sort_wide()
{
	static const int on = env_int("GSR_DEPTH_SORT_9BIT", 1);
	static const int bits = env_int("GSR_DEPTH_SORT_BITS", 3 * RADIX_BITS_WIDE);
	if (!on) return 0;
	// (an odd number of 9-bit passes: 1 .. 9, 19 .. 27 bits)
	return (bits >= 1 && bits <= 27 && (div_up(bits, RADIX_BITS_WIDE) & 1)) ? bits : 3 * RADIX_BITS_WIDE;
}
static thread_local long long t_depth_resorts = 0;   // forward passes of this thread that took the second path (tests)
static bool emit_hist()
{
	static const int env = env_int("GSR_EMIT_HIST", 1);
	return env != 0;
}
// Which binning: depth-first (sort the Gaussians by depth, emit in that order: nine launches for the sort whatever the size) or
// tile-first (compact in id order, sort every tile's list by depth behind the tile sort).  gsr_forward_args.raw_params may force
// either (GSR_BINNING_DEPTH_FIRST / GSR_BINNING_TILE_FIRST); GSR_BINNING=0/1 overrides (the A/B handle); otherwise by size.
constexpr int TILE_FIRST_MAX_GAUSSIANS = 256 * 1024;
static bool binning_tile_first(int raw_params, int P, int tiles)
{
	static const int env = env_int("GSR_BINNING", -1);
	if (env >= 0) return env != 0;
	if (raw_params & GSR_BINNING_TILE_FIRST) return true;
	if (raw_params & GSR_BINNING_DEPTH_FIRST) return false;
	// Measured (profiles/r06_c, r06_d; same box, alternating): the depth sort of the Gaussians is launch-bound up to ~0.5 M visible
	// ones (60 us at 17 k keys, 62 us at 262 k), the per-tile sorts cost ~17-40 ns per thousand instances.  50 k Gaussians @ 640 x 480:
	// the step 0.292 -> 0.259 ms tile-first; 500 k @ 1200 x 680: 0.696 -> 0.686; 2 M @ 1080p: 1.567 -> 1.569; 4 M @ 752 x 480: equal.
	// The instance count is not known when the choice is made: the model's size stands in for it.
	(void)tiles;
	return P <= TILE_FIRST_MAX_GAUSSIANS;
}
static int side_blocks(const gsr_sh_adam* o)
{
	static const int env = env_int("GSR_SH_ADAM_SIDE_BLOCKS", -1);
	return env >= 0 ? env : (o && o->side_blocks > 0 ? o->side_blocks : 256);
}

// Optional per-stage HIP-event timing (gsr_profile_*): events are recorded on the caller's
// stream between the stages of gsr_forward / gsr_backward, so bench.py can price each kernel
// group against its algorithmic bytes without a profiler attached.
// (depth_sort: the depth sort of the Gaussians, or -- tile-first binning -- the count + compaction launches, with offset_scan empty;
// tile_depth_sort: tile-first binning only)
enum { ST_PREPROCESS = 0, ST_DEPTH_SORT, ST_OFFSET_SCAN, ST_EMIT, ST_TILE_SORT, ST_TILE_RANGES, ST_TILE_DEPTH_SORT, ST_BLEND_FWD,
       ST_GRAD_MEMSET, ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_COUNT };
constexpr int ST_FWD_COUNT = ST_BLEND_FWD + 1;
static const char* const k_stage_names[ST_COUNT] = {"preprocess_fwd", "depth_sort", "offset_scan", "emit_instances",
                                                    "tile_sort", "tile_ranges", "tile_depth_sort", "blend_fwd", "grad_memset",
                                                    "blend_bwd", "preprocess_bwd"};
struct Profiler {
	std::atomic<int> on{0};   // 0 off, 1 every stage, 2 only the backward blend (its two events: an event record costs a
	                          // ~5 us pipeline bubble, eleven of them 2 % of a C3 train step)
	hipEvent_t fwd[ST_FWD_COUNT + 1] = {};   // boundaries of the forward stages
	hipEvent_t bwd[4] = {};   // boundaries of the 3 backward stages
	// (written by the thread that runs the pass -- PyTorch's autograd worker for the backward pass -- and read by the thread that
	// asks for the timings: atomics; the event handles are created once, before `on` is set)
	std::atomic<bool> created{false}, fwd_done{false}, bwd_done{false};
	int create()
	{
		if (created) return GSR_OK;
		for (auto& e : fwd) GSR_HIP(hipEventCreate(&e));
		for (auto& e : bwd) GSR_HIP(hipEventCreate(&e));
		created = true;
		return GSR_OK;
	}
};
// process-wide (not per thread): PyTorch runs backward on an autograd worker thread, and the
// benchmark reads the timings from the main thread.  Intended for single-stream benchmarking.
static Profiler t_prof;
#define PROF_FWD(i) do { if (t_prof.on == 1) GSR_HIP(hipEventRecord(t_prof.fwd[i], stream)); } while (0)
#define PROF_BWD(i) do { if (t_prof.on == 1 || (t_prof.on == 2 && ((i) == 1 || (i) == 2))) GSR_HIP(hipEventRecord(t_prof.bwd[i], stream)); } while (0)

static inline size_t geometry_bytes(int P)
{
	size_t b = 0;
	GeometryState::carve(nullptr, (size_t)P, &b);
	return b;
}
static inline size_t binning_bytes(int R)
{
	size_t b = 0;
	BinningState::carve(nullptr, (size_t)R, &b);
	return b;
}
static inline size_t image_bytes(int W, int H)
{
	size_t b = 0;
	const size_t T = (size_t)div_up(W, TILE) * div_up(H, TILE);
	ImageState::carve(nullptr, (size_t)W * H, T, &b);
	return b;
}

static int validate_common(int P, int D, int M, int W, int H, const void* shs, const void* colors, const void* scales,
                           const void* rotations, const void* cov3D)
{
	if (P < 0 || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARG;
	if ((shs == nullptr) == (colors == nullptr)) return GSR_ERR_INVALID_ARG;  // exactly one (gaussian_rasterizer.cpp:201-203)
	const bool sr = scales != nullptr && rotations != nullptr;
	if ((scales != nullptr) != (rotations != nullptr)) return GSR_ERR_INVALID_ARG;
	if (sr == (cov3D != nullptr)) return GSR_ERR_INVALID_ARG;                 // exactly one (:205-207)
	if (shs) {
		if (D < 0 || D > 3) return GSR_ERR_UNSUPPORTED;
		if (M < (D + 1) * (D + 1)) return GSR_ERR_INVALID_ARG;
	}
	if (div_up(W, TILE) > 65535 || div_up(H, TILE) > 65535) return GSR_ERR_UNSUPPORTED;
	return GSR_OK;
}

// gsr_sh_adam with ->lazy set -> the device-side description: the scalars of Adam step (step - k) at [k], formed exactly as
// adam_scalars forms them for the eager step.  shs: the tensor the rasterizer call works on (must be o.param).
static int make_lazy_adam(const gsr_sh_adam& o, const float* shs, int M, LazyAdam& la)
{
	const gsr_sh_adam_lazy& z = *o.lazy;
	if (!o.param || (shs && o.param != shs) || !o.exp_avg || !o.exp_avg_sq || !z.row_step || o.step < 1 || z.window < 2 ||
	    z.window > GSR_SH_LAZY_WINDOW)
		return GSR_ERR_INVALID_ARG;
	if (M != 16 || ((reinterpret_cast<uintptr_t>(o.param) | reinterpret_cast<uintptr_t>(o.exp_avg) | reinterpret_cast<uintptr_t>(o.exp_avg_sq)) & 15))
		return GSR_ERR_UNSUPPORTED;
	static_assert(GSR_SH_LAZY_WINDOW == LAZY_WINDOW_MAX, "gsr.h and kernels.h agree on the window");
	la.param = o.param; la.exp_avg = o.exp_avg; la.exp_avg_sq = o.exp_avg_sq;
	la.row_step = z.row_step; la.step = o.step; la.window = z.window;
	for (int k = 0; k < LAZY_WINDOW_MAX; k++) {
		AdamScalars s{};
		if (k < z.window && o.step - k >= 1)
			s = adam_scalars(k == 0 ? o.lr : z.lr_past[k - 1], k == 0 ? o.lr_tail : z.lr_tail_past[k - 1], o.beta1, o.beta2, o.eps, o.step - k);
		la.t.step_size[k] = s.step_size; la.t.step_size_tail[k] = s.step_size_tail; la.t.inv_sqrt_bc2[k] = s.inv_sqrt_bc2;
	}
	const AdamScalars c = adam_scalars(o.lr, o.lr_tail, o.beta1, o.beta2, o.eps, o.step);
	la.t.b1 = c.b1; la.t.b2 = c.b2; la.t.omb1 = c.omb1; la.t.omb2 = c.omb2; la.t.eps = c.eps;
	return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_geometry_bytes(int P) { return geometry_bytes(P < 0 ? 0 : P); }
size_t gsr_binning_bytes(int R) { return binning_bytes(R < 0 ? 0 : R); }
size_t gsr_image_bytes(int W, int H) { return (W <= 0 || H <= 0) ? 0 : image_bytes(W, H); }
size_t gsr_knn_scratch_bytes(int P) { return knn_scratch_bytes(P < 0 ? 0 : P); }

const char* gsr_strerror(int status)
{
	switch (status) {
		case GSR_OK: return "ok";
		case GSR_ERR_INVALID_ARG: return "invalid argument";
		case GSR_ERR_ALLOC: return "scratch allocation callback returned NULL";
		case GSR_ERR_HIP: return "HIP runtime error";
		case GSR_ERR_UNSUPPORTED: return "unsupported configuration";
		default: return "unknown status";
	}
}
int gsr_last_hip_error(void) { return t_last_err; }
const char* gsr_last_hip_error_string(void) { return t_last_what; }
const char* gsr_backend(void)
{
#ifdef GSR_EMU
	return "emu-wave64";
#else
	return "hip-gfx950";
#endif
}

int gsr_forward(const gsr_forward_args* a, gsr_alloc_fn geometryBuffer, void* geometry_ctx, gsr_alloc_fn binningBuffer,
                void* binning_ctx, gsr_alloc_fn imageBuffer, void* image_ctx, void* stream_, int* num_rendered)
{
	if (!a || !geometryBuffer || !binningBuffer || !imageBuffer || !num_rendered) return GSR_ERR_INVALID_ARG;
	*num_rendered = 0;
	int st = validate_common(a->P, a->D, a->M, a->width, a->height, a->shs, a->colors_precomp, a->scales, a->rotations,
	                         a->cov3D_precomp);
	if (a->P == 0) {   // src/rasterize_points.cu:81
		t_last_visible = 0;   // (an empty model sees nothing: gsr_last_visible_count() must not keep the previous view's count)
		return GSR_OK;
	}
	if (st != GSR_OK) return st;
	if (!a->background || !a->means3D || !a->opacities || !a->viewmatrix || !a->projmatrix || !a->cam_pos || !a->out_color)
		return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	const int P = a->P, W = a->width, H = a->height;
	const int grid_x = div_up(W, TILE), grid_y = div_up(H, TILE), tiles = grid_x * grid_y;

	char* geom_chunk = geometryBuffer(geometry_ctx, geometry_bytes(P));
	if (!geom_chunk) return GSR_ERR_ALLOC;
	GeometryState g = GeometryState::carve(geom_chunk, (size_t)P);
	char* img_chunk = imageBuffer(image_ctx, image_bytes(W, H));
	if (!img_chunk) return GSR_ERR_ALLOC;
	ImageState im = ImageState::carve(img_chunk, (size_t)W * H, (size_t)tiles);

	HostSync* sync_ = nullptr;
	if ((st = host_sync(&sync_)) != GSR_OK) return st;   // (GSR_ERR_UNSUPPORTED: a device ordinal beyond MAX_DEVICES)
	HostSync& t_sync = *sync_;
	if ((st = t_sync.init()) != GSR_OK) return st;
	t_prof.fwd_done = false;
	// (nothing to zero: the projection kernel's waves leave their counts with plain stores -- state.h: wave_counts)
	PROF_FWD(0);

	PreprocessParams pp;
	pp.P = P; pp.D = a->D; pp.M = a->M;
	pp.means3D = a->means3D; pp.scales = a->scales; pp.scale_modifier = a->scale_modifier; pp.rotations = a->rotations;
	pp.opacities = a->opacities; pp.shs = a->shs; pp.cov3D_precomp = a->cov3D_precomp; pp.colors_precomp = a->colors_precomp;
	pp.view = a->viewmatrix; pp.proj = a->projmatrix; pp.campos = a->cam_pos;
	pp.W = W; pp.H = H; pp.tan_fovx = a->tan_fovx; pp.tan_fovy = a->tan_fovy;
	pp.focal_y = H / (2.0f * a->tan_fovy);  // rasterizer_impl.cu:221-222
	pp.focal_x = W / (2.0f * a->tan_fovx);
	pp.grid_x = grid_x; pp.grid_y = grid_y; pp.radii_out = a->radii;
	pp.raw_params = a->raw_params | (cov3D_stored() ? GSR_STORE_COV3D : 0);
	pp.ranges = im.ranges; pp.tiles = tiles;   // zeroed there: rasterizer_impl.cu:310
	pp.lazy = LazyAdam{};
	if (a->sh_adam && a->sh_adam->lazy) {   // lazy SH Adam: visible rows that lag behind take their missed steps first
		if (!a->shs) return GSR_ERR_INVALID_ARG;
		if ((st = make_lazy_adam(*a->sh_adam, a->shs, a->M, pp.lazy)) != GSR_OK) return st;
	}
	if ((st = launch_preprocess_fwd(pp, g, stream)) != GSR_OK) return st;

	PROF_FWD(1);
	// num_rendered and the visible count reach the host without a copy in the stream: the depth sort's first two launches add up
	// the waves' pairs on the side (sort.hip) and store the totals into the mapped pinned words; the event is recorded behind them
	RadixHostCount hc;
	hc.pairs = g.wave_counts; hc.n = (int)wave_count_slots((size_t)P); hc.partials = g.count_partials; hc.host_out = t_sync.pinned_dev; hc.ready = t_sync.ev;

	// depth order (stable: equal depths keep ascending Gaussian id).  The first pass reads all P keys and drops the culled
	// Gaussians (key 0xFFFFFFFF, RADIX_INVALID_KEY), leaving V in g.visible; the other three passes and the scan run over the
	// V visible ones only (V = 0.47 P at C3).  order[V..P) is undefined, offsets[V..P) = R: the culled Gaussians used to sort
	// to the end with exactly that offset, so the instance emission sees the same arrays.
	//
	// Three passes, not four: a visible Gaussian has z > 0.2 (the frustum test), so its key -- the bits of a positive float --
	// exceeds DEPTH_KEY_BIAS = bits(0.2f), and key - DEPTH_KEY_BIAS (an order-preserving shift) has its high five bits zero for
	// every z < 0.2 * 2^16 = 13 107: 27 bits = 3 digits of 9.  The largest key of the view reaches the host with the instance
	// count; a view that holds a Gaussian beyond that range (or a NaN depth) sorts again with the plain four passes of eight
	// bits (depth_sort_wide()/GSR_DEPTH_SORT_9BIT=0: the A/B handle; GSR_DEPTH_SORT_BITS: the test handle of the second path).
	uint32_t *kres = nullptr, *vres = nullptr;
	const int narrow_bits = depth_sort_wide();   // 0 = the plain sort
	auto plain_depth_sort = [&](const RadixHostCount* ride) {
		return launch_radix_sort(g.depth_key, nullptr, g.sort_keys_a, g.order, g.sort_keys_b, g.sort_vals_b, P, 0, 32, g.sort_scratch, stream,
		                         &kres, &vres, g.visible, ride);   // (4 passes end in the ping buffers: vres == g.order)
	};
	auto offset_scan = [&]() {
		// (g.sort_keys_b, the depth sort's spare buffer, is free again: it receives the emission's seeds -- binning.hip)
		return launch_scan_rect_tiles(reinterpret_cast<const uint2*>(g.rect), g.order, g.offsets, g.rect_sorted, P, g.scan_scratch, stream,
		                              g.visible, g.sort_keys_b, EMIT_SEED_STRIDE, (uint32_t)P, g.long_runs, g.long_counts, g.long_capacity);
	};
	const bool tile_first = binning_tile_first(a->raw_params, P, tiles);
	if (tile_first) {
		// no order among the Gaussians: the visible ones compacted by ascending id (g.order), their offsets, rectangles, the emission's
		// seeds and the list of long runs in ONE pass; the counts reach the host from a one-workgroup launch in front of it
		st = launch_compact_visible(g.tiles_touched, reinterpret_cast<const uint2*>(g.rect), g.wave_counts, hc.n, g.count_partials, g.order, g.offsets,
		                            g.rect_sorted, P, t_sync.pinned_dev, t_sync.ev, stream, g.sort_keys_b, EMIT_SEED_STRIDE, (uint32_t)P, g.long_runs,
		                            g.long_counts, g.long_capacity, g.visible);
		if (st != GSR_OK) return st;
		PROF_FWD(2);
	} else {
		if (narrow_bits) {
			// (ping and pong swapped: an odd number of passes ends in the pong buffers, and the order must end in g.order)
			st = launch_radix_sort(g.depth_key, nullptr, g.sort_keys_b, g.sort_vals_b, g.sort_keys_a, g.order, P, 0, narrow_bits, g.sort_scratch, stream,
			                       &kres, &vres, g.visible, &hc, false, RADIX_BITS_WIDE, DEPTH_KEY_BIAS);
			if (st == GSR_OK && vres != g.order) st = GSR_ERR_INVALID_ARG;   // (an even number of wide passes: not a configuration of this library)
		} else
			st = plain_depth_sort(&hc);
		if (st != GSR_OK) return st;
		PROF_FWD(2);
		if ((st = offset_scan()) != GSR_OK) return st;
	}
	PROF_FWD(3);

	{
		const auto w0 = std::chrono::steady_clock::now();
		GSR_HIP(hipEventSynchronize(t_sync.ev));
		t_sync_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
		t_sync_waits++;
	}
	const unsigned long long R64 = (unsigned long long)t_sync.pinned[0] | ((unsigned long long)t_sync.pinned[1] << 32);
	const unsigned long long V64 = t_sync.pinned[2];
	t_last_visible = (int)V64;
	if (!tile_first && narrow_bits && V64 != 0 && (t_sync.pinned[3] - DEPTH_KEY_BIAS) >> narrow_bits) {
		// a depth beyond the three-pass range (z >= 13 107, or not a number): the order just produced is wrong for those keys --
		// sort again on all 32 bits and redo the offsets (the list of long runs is built by the scan: its counters start over)
		GSR_HIP(hipMemsetAsync(g.long_counts, 0, (size_t)LONG_LISTS * LONG_COUNT_STRIDE * sizeof(uint32_t), stream));
		if ((st = plain_depth_sort(nullptr)) != GSR_OK) return st;
		if ((st = offset_scan()) != GSR_OK) return st;
		t_depth_resorts++;
	}
	if (R64 > 0x7FFFFFFFull) return GSR_ERR_UNSUPPORTED;  // more than 2^31 instances
	const int R = (int)R64;
	char* bin_chunk = binningBuffer(binning_ctx, binning_bytes(R));
	if (!bin_chunk) return GSR_ERR_ALLOC;
	BinningState bs = BinningState::carve(bin_chunk, (size_t)R);

	// (im.ranges: zeroed by preprocess_fwd)
	uint32_t* point_list = bs.vals_a;
	if (R > 0) {
		const int bits = (int)higher_msb((uint32_t)tiles);
		// GSR_CULL_EMPTY_TILES: instances of tiles in which no pixel can blend the Gaussian leave the list in the tile sort's
		// first pass (78 % of the rectangle instances at C3 blend into no pixel); the count that remains lives on the device
		// (a sort of zero passes -- a one-tile image -- cannot drop anything: the flag is ignored there)
		const bool cull = (a->raw_params & GSR_CULL_EMPTY_TILES) != 0 && bits > 0;
		uint32_t* const listed = cull ? g.visible + 16 : nullptr;
		// (the emission counts the tile sort's first histogram on the way: GSR_EMIT_HIST=0 is the A/B handle for the separate launch)
		// (a small list on a grid of up to 2 048 tiles: all tile bits in one pass, the ranges from that pass -- state.h: tile_sort_digit_bits)
		const int digit_bits = tile_sort_digit_bits(tiles, R);
		const bool one_pass = digit_bits == RADIX_BITS_ONE_PASS;
		const int hist_bits = emit_hist() ? radix_first_pass_bits(0, bits, digit_bits) : 0;
		// (tile-first: the compacted arrays hold V entries and nothing behind them -- the emission is told so)
		if ((st = launch_emit_instances(tile_first ? (int)V64 : P, R, g, grid_x, bs.keys_a, bs.vals_a, bs.touched, stream, cull ? 1 : 0, emit_seeded(), bs.sort_scratch, hist_bits)) != GSR_OK) return st;
		PROF_FWD(4);
		uint32_t* tkeys = nullptr;
		if ((st = launch_radix_sort(bs.keys_a, bs.vals_a, bs.keys_a, bs.vals_a, bs.keys_b, bs.vals_b, R, 0, bits,
		                            bs.sort_scratch, stream, &tkeys, &point_list, listed, nullptr, hist_bits > 0, digit_bits, 0u,
		                            one_pass ? im.ranges : (uint2*)nullptr)) != GSR_OK)
			return st;
		PROF_FWD(5);
		if (!one_pass && (st = launch_tile_ranges(R, tkeys, im.ranges, stream, listed)) != GSR_OK) return st;
		PROF_FWD(6);
		if (tile_first) {
			// the instances reached their tiles in id order: every tile's list by depth now (equal depths keep ascending id).  Scratch:
			// the tile sort's spare pair and the forward blend's flag planes (4 R bytes, written only by the blend behind this)
			const bool in_a = point_list == bs.vals_a;
			if ((st = launch_tile_depth_sort(im.ranges, tiles, g.depth_key, point_list, in_a ? bs.keys_b : bs.keys_a, in_a ? bs.vals_b : bs.vals_a,
			                                 reinterpret_cast<uint32_t*>(bs.contrib), stream)) != GSR_OK)
				return st;
		}
	} else {
		PROF_FWD(4);
		PROF_FWD(5);
		PROF_FWD(6);
	}
	PROF_FWD(7);

	BlendFwdParams bp;
	bp.ranges = im.ranges; bp.point_list = point_list; bp.rec = g.rec; bp.bg = a->background;
	bp.final_T = im.final_T; bp.n_contrib = im.n_contrib; bp.out_color = a->out_color;
	bp.contrib = bs.contrib; bp.contrib_stride = (size_t)R;
	bp.W = W; bp.H = H; bp.grid_x = grid_x; bp.tiles = tiles;
	bp.deal = make_tile_deal(tiles, grid_x, xcd_deal_mode(tiles));
	if ((st = launch_blend_fwd(bp, stream)) != GSR_OK) return st;
	PROF_FWD(8);
	t_prof.fwd_done = t_prof.on == 1;
	*num_rendered = R;
	return GSR_OK;
}

int gsr_backward(const gsr_backward_args* a, void* stream_)
{
	if (!a) return GSR_ERR_INVALID_ARG;
	int st = validate_common(a->P, a->D, a->M, a->width, a->height, a->shs, a->colors_precomp, a->scales, a->rotations,
	                         a->cov3D_precomp);
	if (a->P == 0) return GSR_OK;
	if (st != GSR_OK) return st;
	if (!a->background || !a->means3D || !a->viewmatrix || !a->projmatrix || !a->campos || !a->geom_buffer ||
	    !a->image_buffer || !a->dL_dpix || !a->dL_dcolor || !a->dL_dmean3D || a->R < 0)
		return GSR_ERR_INVALID_ARG;
	if (!a->dL_dopacity && !a->geom_adam) return GSR_ERR_INVALID_ARG;   // (dL_dmean2D / dL_dcov3D: nullable, see gsr.h)
	if (a->shs && !a->dL_dsh && !a->dL_dcolor_view && !a->sh_adam) return GSR_ERR_INVALID_ARG;
	if ((a->stat_grad_accum != nullptr) != (a->stat_denom != nullptr) || (a->stat_denom != nullptr) != (a->stat_max_radii != nullptr))
		return GSR_ERR_INVALID_ARG;
	// (sh_adam together with dL_dcolor_view: only its lazy form, which then means "catch up this step's slice ahead of the
	// exchange", see gsr.h)
	if (a->sh_adam && (!a->shs || (a->dL_dcolor_view && !a->sh_adam->lazy) || !a->sh_adam->exp_avg || !a->sh_adam->exp_avg_sq ||
	                   a->sh_adam->step < 1))
		return GSR_ERR_INVALID_ARG;
	if (a->dL_dcolor_view && !a->shs) return GSR_ERR_INVALID_ARG;
	if (a->scales && (!a->dL_dscale || !a->dL_drot) && !a->geom_adam) return GSR_ERR_INVALID_ARG;
	if (a->R > 0 && !a->binning_buffer) return GSR_ERR_INVALID_ARG;
	// (a stream cannot be made to wait for an event of its own future: refused HERE, before anything is enqueued and before the
	// lazy rows' catch-up has advanced a step counter)
	if (a->color_view_ready_stream && a->dL_dcolor_view && a->color_view_ready_stream == stream_) return GSR_ERR_INVALID_ARG;
	if (a->packed_view && a->dL_dcolor_view && (a->packed_capacity_rows < 0 || (a->packed_capacity_rows & 3))) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	const int P = a->P, W = a->width, H = a->height, R = a->R;
	const int grid_x = div_up(W, TILE), grid_y = div_up(H, TILE), tiles = grid_x * grid_y;
	GeometryState g = GeometryState::carve(a->geom_buffer, (size_t)P);
	ImageState im = ImageState::carve(a->image_buffer, (size_t)W * H, (size_t)tiles);
	BinningState bs = BinningState::carve(a->binning_buffer, (size_t)R);
	const int passes = tile_sort_passes(tiles, R);
	const uint32_t* point_list = (passes % 2) ? bs.vals_b : bs.vals_a;

	HostSync* sync_ = nullptr;
	if (gsr_status hs = host_sync(&sync_); hs != GSR_OK) return hs;
	HostSync& t_sync = *sync_;
	t_prof.bwd_done = false;
	// ---- the rest of the validation, BEFORE anything is enqueued: a call that is going to be refused must not have applied a
	// part of an optimizer step already (the culled rows' update below is forked onto a second stream first thing)
	const bool pre_slice = a->sh_adam && a->dL_dcolor_view;   // factored mode: no fused step, only the lazy rows' catch-up
	const bool rows_path = sh_rows_path(a->shs, a->M, a->D, a->dL_dcolor_view != nullptr, a->sh_adam != nullptr && !pre_slice, a->dL_dsh);
	if (!a->dL_dcov3D && a->cov3D_precomp) return GSR_ERR_INVALID_ARG;
	if (a->sh_adam) {
		const gsr_sh_adam& o = *a->sh_adam;
		if (o.param != a->shs || !o.param) return GSR_ERR_INVALID_ARG;   // the writable alias of the (const) SH input
		// the fused step exists for aligned [P,16,3] rows only (launch_preprocess_bwd)
		if (!rows_path || ((reinterpret_cast<uintptr_t>(o.exp_avg) | reinterpret_cast<uintptr_t>(o.exp_avg_sq)) & 15))
			return GSR_ERR_UNSUPPORTED;
	}
	GeomAdam geom{};
	if (a->geom_adam) {
		const gsr_geom_adam& o = *a->geom_adam;
		const int all_raw = GSR_RAW_OPACITY | GSR_RAW_SCALING | GSR_RAW_ROTATION;
		const gsr_adam_tensor* ts[4] = {&o.xyz, &o.opacity, &o.scaling, &o.rotation};
		for (const gsr_adam_tensor* t : ts)
			if (!t->param || !t->exp_avg || !t->exp_avg_sq || t->step < 1) return GSR_ERR_INVALID_ARG;
		if ((a->raw_params & all_raw) != all_raw || a->cov3D_precomp || o.xyz.param != a->means3D || o.scaling.param != a->scales ||
		    o.rotation.param != a->rotations || !a->dL_dmean3D || a->dL_dcolor_view)
			return GSR_ERR_INVALID_ARG;
		if ((reinterpret_cast<uintptr_t>(o.rotation.param) | reinterpret_cast<uintptr_t>(o.rotation.exp_avg) |
		     reinterpret_cast<uintptr_t>(o.rotation.exp_avg_sq)) & 15)
			return GSR_ERR_UNSUPPORTED;
		// the fused geometry step lives in the two-kernel path of the reference's SH layout (launch_preprocess_bwd)
		if (!(rows_path && a->scales && a->rotations)) return GSR_ERR_UNSUPPORTED;
		auto fill = [&](const gsr_adam_tensor& t, GeomAdamTensor& gt) {
			gt.param = t.param; gt.exp_avg = t.exp_avg; gt.exp_avg_sq = t.exp_avg_sq;
			gt.s = adam_scalars(t.lr, t.lr, o.beta1, o.beta2, o.eps, t.step);
		};
		geom.on = 1;
		fill(o.xyz, geom.xyz); fill(o.opacity, geom.opacity); fill(o.scaling, geom.scaling); fill(o.rotation, geom.rotation);
	}
	// Fused Adam step of the SH tensor: the rows of the CULLED Gaussians carry a zero gradient, i.e. their update does not
	// depend on anything this pass computes.  It runs on a second stream from here on -- HBM-bound streaming (1152 B per culled
	// Gaussian) next to the VALU-bound backward blend, which leaves HBM nearly idle -- and is joined at the end; the row kernel
	// behind preprocess_bwd then updates the visible rows only.  (Only the radii of the forward pass are read.)
	bool side_busy = false;
	// an error exit taken while the second stream holds work of this call: join it first, so that nothing of this call is still
	// writing the caller's tensors when the caller sees the status
	auto fail = [&](int status) -> int {
		if (side_busy) (void)hipStreamWaitEvent(stream, t_sync.join, 0);
		return status;
	};
	const bool lazy = a->sh_adam && a->sh_adam->lazy;
	LazyAdam la{};
	if (lazy) {
		// lazy mode (gsr_sh_adam_lazy): the culled rows do NOT take this step now; a rotating 1/window of the row blocks catches
		// up instead (forked next to the backward blend, below)
		if (!a->shs) return GSR_ERR_INVALID_ARG;
		if ((st = make_lazy_adam(*a->sh_adam, a->shs, a->M, la)) != GSR_OK) return st;
		if (pre_slice && la.window < 3) return GSR_ERR_INVALID_ARG;
	}
	// This step's slice of the lazy rows: 1/window of the culled rows, each taking `window` zero-gradient steps in registers --
	// little traffic (72 MB at C3), mostly arithmetic -- on the second stream if there is one; its rows are disjoint from the
	// visible ones the per-Gaussian backward kernels update.
	auto launch_lazy_slice = [&]() -> int {
		const int* radii = pre_slice ? nullptr : (a->radii ? a->radii : g.radii);
		const int mode = pre_slice ? 3 : 0;
		if (!side_stream_enabled(a->sh_adam)) return launch_sh_adam_lazy(P, radii, la, stream, mode);
		int s2 = t_sync.init_side();
		if (s2 != GSR_OK) return s2;
		GSR_HIP(hipEventRecord(t_sync.fork, stream));
		GSR_HIP(hipStreamWaitEvent(t_sync.side, t_sync.fork, 0));
		s2 = launch_sh_adam_lazy(P, radii, la, t_sync.side, mode);
		// (recorded even after a failed launch: whatever did reach the second stream is joined by fail())
		GSR_HIP(hipEventRecord(t_sync.join, t_sync.side));
		side_busy = true;
		return s2;
	};
	if (lazy) {
		// (forked next to the backward blend, below)
	} else if (a->sh_adam && a->M == 16 && a->D >= 0 && a->D <= 3 && side_stream_enabled(a->sh_adam)) {
		const gsr_sh_adam& o = *a->sh_adam;
		if ((st = t_sync.init_side()) != GSR_OK) return st;
		const RowAdam ra = {o.param, o.exp_avg, o.exp_avg_sq, adam_scalars(o.lr, o.lr_tail, o.beta1, o.beta2, o.eps, o.step)};
		GSR_HIP(hipEventRecord(t_sync.fork, stream));
		GSR_HIP(hipStreamWaitEvent(t_sync.side, t_sync.fork, 0));
		st = launch_sh_adam_culled(P, a->radii ? a->radii : g.radii, ra, t_sync.side, side_blocks(a->sh_adam));
		GSR_HIP(hipEventRecord(t_sync.join, t_sync.side));
		side_busy = true;
		if (st != GSR_OK) return fail(st);
	}
	// The lazy rows' slice is forked HERE, next to the backward blend (gsr_sh_adam.lazy_slice_late: behind the blend, next to the
	// per-Gaussian kernels, as until r03_r).  Round 2 measured the two placements equal; since the fused SH step keeps its
	// parameter rows in LDS (16 instead of 24 waves per CU) it is the one that suffers from a neighbour: same box, C3 1.660 ->
	// 1.637 ms, a C5 view 1.768 -> 1.738 (the blend pays 5-13 us, the per-Gaussian stage gains 30-40).
	const bool slice_early = lazy_slice_early(a->sh_adam);
	if (lazy && slice_early && (st = launch_lazy_slice()) != GSR_OK) return fail(st);
	PROF_BWD(0);
	// per-instance gradient slots of the blend backward (48 B/instance, inside the binning buffer);
	// every API output is written exactly once by preprocess_bwd
	// R bytes of flags instead of 48 R bytes of slots (+ the 64 pad bytes: the reader's byte->bit squeeze needs every byte 0/1)
	// The forward pass hands the flags over cleared (emit_instances_kernel), and on the usual path -- aligned [P,16,3] SH rows:
	// sh_bwd_rows_kernel runs last -- this pass leaves them cleared again; only the other paths clear them here.
	if (R > 0 && !rows_path) GSR_HIP(hipMemsetAsync(bs.touched, 0, touched_clear_bytes((size_t)R), stream));
	PROF_BWD(1);
	if (R > 0) {
		BlendBwdParams bp;
		bp.ranges = im.ranges; bp.point_list = point_list; bp.rec = g.rec; bp.bg = a->background;
		bp.final_T = im.final_T; bp.n_contrib = im.n_contrib; bp.dL_dpix = a->dL_dpix;
		bp.partials = bs.partials;
		bp.touched = bs.touched;
		bp.contrib = bs.contrib; bp.contrib_stride = (size_t)R;
		bp.W = W; bp.H = H; bp.grid_x = grid_x; bp.tiles = tiles;
		bp.deal = make_tile_deal(tiles, grid_x, xcd_deal_mode(tiles));
		if ((st = launch_blend_bwd(bp, stream)) != GSR_OK) return fail(st);
	}
	PROF_BWD(2);
	if (lazy && !slice_early && (st = launch_lazy_slice()) != GSR_OK) return fail(st);

	PreprocessBwdParams pb;
	pb.P = P; pb.D = a->D; pb.M = a->M;
	pb.means3D = a->means3D; pb.radii = a->radii ? a->radii : g.radii; pb.shs = a->shs; pb.clamped = g.clamped;
	pb.scales = a->scales; pb.rotations = a->rotations; pb.scale_modifier = a->scale_modifier;
	pb.cov3D = a->cov3D_precomp;   // (null: preprocess_bwd recomputes the covariance from scales / rotations, kernels.h)
	if (!pb.cov3D && cov3D_stored()) pb.cov3D = g.cov3D;
	pb.view = a->viewmatrix; pb.proj = a->projmatrix; pb.campos = a->campos;
	pb.focal_y = H / (2.0f * a->tan_fovy);
	pb.focal_x = W / (2.0f * a->tan_fovx);
	pb.tan_fovx = a->tan_fovx; pb.tan_fovy = a->tan_fovy;
	pb.tiles_touched = g.tiles_touched; pb.partials = R > 0 ? bs.partials : nullptr; pb.touched = R > 0 ? bs.touched : nullptr;
	pb.long_runs = g.long_runs; pb.long_counts = g.long_counts; pb.long_capacity = g.long_capacity;
	pb.half_w = 0.5f * (float)W; pb.half_h = 0.5f * (float)H;
	pb.rec = g.rec; pb.raw_params = a->raw_params;
	pb.dL_dmean2D = a->dL_dmean2D; pb.dL_dconic = a->dL_dconic; pb.dL_dopacity = a->dL_dopacity; pb.dL_dcolor = a->dL_dcolor;
	pb.dL_dmean3D = a->dL_dmean3D; pb.dL_dcov3D = a->dL_dcov3D; pb.dL_dsh = a->dL_dsh; pb.dL_dscale = a->dL_dscale;
	pb.dL_drot = a->dL_drot;
	pb.dL_dcolor_view = a->dL_dcolor_view;
	pb.packed_msg = a->dL_dcolor_view ? a->packed_view : nullptr;
	pb.packed_capacity = a->packed_capacity_rows;
	pb.stat_accum = a->stat_grad_accum; pb.stat_denom = a->stat_denom; pb.stat_max_radii = a->stat_max_radii;
	pb.adam_param = nullptr; pb.adam_exp_avg = nullptr; pb.adam_exp_avg_sq = nullptr;
	pb.adam = AdamScalars{};
	pb.adam_skip_culled = 0;
	pb.lazy_row_step = nullptr; pb.lazy_step = 0;
	pb.touched_clear = (R > 0 && rows_path) ? bs.touched : nullptr;
	pb.touched_clear_bytes = (uint32_t)touched_clear_bytes((size_t)R);
	if (a->sh_adam && !pre_slice) {
		const gsr_sh_adam& o = *a->sh_adam;   // the same scalars gsr_adam_step derives (kernels.h: adam_scalars)
		pb.adam_param = o.param;
		pb.adam_exp_avg = o.exp_avg; pb.adam_exp_avg_sq = o.exp_avg_sq;
		pb.adam = adam_scalars(o.lr, o.lr_tail, o.beta1, o.beta2, o.eps, o.step);
		pb.adam_skip_culled = (side_busy || lazy) ? 1 : 0;
		if (lazy) { pb.lazy_row_step = la.row_step; pb.lazy_step = la.step; }
	}
	pb.geom = geom;
	pb.notify_stream = nullptr; pb.notify_event = nullptr;
	if (a->color_view_ready_stream && a->dL_dcolor_view) {
		if ((st = t_sync.init_notify()) != GSR_OK) return fail(st);
		pb.notify_stream = a->color_view_ready_stream;
		pb.notify_event = (void*)t_sync.notify;
	}
	if ((st = launch_preprocess_bwd(pb, stream)) != GSR_OK) return fail(st);
	if (side_busy) GSR_HIP(hipStreamWaitEvent(stream, t_sync.join, 0));   // whatever follows on the caller's stream sees the whole update
	PROF_BWD(3);
	t_prof.bwd_done = t_prof.on != 0;
	return GSR_OK;
}

int gsr_sh_adam_flush(int P, const gsr_sh_adam* adam, void* stream_)
{
	if (P < 0 || !adam || !adam->lazy) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	LazyAdam la{};
	int st = make_lazy_adam(*adam, nullptr, 16, la);
	if (st != GSR_OK) return st;
	return launch_sh_adam_lazy(P, nullptr, la, (hipStream_t)stream_, 1);
}

int gsr_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                           long long campos_stride, const float* dL_dcolor_views, long long view_stride, float scale,
                           float* dL_dsh, void* stream_)
{
	if (P < 0 || n_views < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || campos_stride < 0 || view_stride < 0)
		return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!means3D || !campos || !dL_dcolor_views || !dL_dsh) return GSR_ERR_INVALID_ARG;
	return launch_sh_grad_from_views(P, D, M, n_views, means3D, campos, campos_stride, dL_dcolor_views, view_stride, scale,
	                                 dL_dsh, nullptr, (hipStream_t)stream_);
}

int gsr_sh_adam_from_views(int P, int D, int M, int n_views, const float* means3D, const float* campos,
                           long long campos_stride, const float* dL_dcolor_views, long long view_stride, float scale,
                           float* shs, const gsr_sh_adam* o, void* stream_)
{
	if (P < 0 || n_views < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || campos_stride < 0 || view_stride < 0)
		return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!means3D || !campos || !dL_dcolor_views || !shs || !o || !o->exp_avg || !o->exp_avg_sq || o->step < 1)
		return GSR_ERR_INVALID_ARG;
	if (o->param && o->param != shs) return GSR_ERR_INVALID_ARG;
	const RowAdam ra = {shs, o->exp_avg, o->exp_avg_sq, adam_scalars(o->lr, o->lr_tail, o->beta1, o->beta2, o->eps, o->step)};
	LazyAdam la{};
	if (o->lazy) {
		gsr_sh_adam with_param = *o;
		with_param.param = shs;
		const int st = make_lazy_adam(with_param, shs, M, la);
		if (st != GSR_OK) return st;
	}
	return launch_sh_grad_from_views(P, D, M, n_views, means3D, campos, campos_stride, dL_dcolor_views, view_stride, scale,
	                                 nullptr, &ra, (hipStream_t)stream_, o->lazy ? &la : nullptr);
}

size_t gsr_packed_view_words(int P, int capacity_rows) { return (P < 0 || capacity_rows < 0) ? 0 : packed_view_words(P, capacity_rows); }
size_t gsr_pack_scratch_bytes(int P) { return P <= 0 ? 0 : (scan_scratch_elems((int)pack_groups(P)) + 64) * sizeof(uint32_t); }

int gsr_pack_view_plan(int P, const int* radii, uint32_t* message, void* scratch, void* stream_)
{
	if (P < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!radii || !message || !scratch) return GSR_ERR_INVALID_ARG;
	return launch_pack_view_plan(P, radii, message, static_cast<uint32_t*>(scratch), (hipStream_t)stream_);
}

int gsr_pack_color_view(int P, const float* dL_dcolor_view, const float* campos, int capacity_rows, uint32_t* message, void* scratch,
                        void* stream_)
{
	if (P < 0 || capacity_rows < 0 || (capacity_rows & 3)) return GSR_ERR_INVALID_ARG;   // (a multiple of 4 rows: the messages stay 16-byte aligned)
	if (P == 0) return GSR_OK;
	if (!dL_dcolor_view || !campos || !message || !scratch) return GSR_ERR_INVALID_ARG;
	return launch_pack_color_view(P, dL_dcolor_view, campos, capacity_rows, message, static_cast<uint32_t*>(scratch), (hipStream_t)stream_);
}

int gsr_sh_grad_from_packed_views(int P, int D, int M, int n_views, const float* means3D, const uint32_t* messages, long long msg_stride,
                                  float scale, float* dL_dsh, void* stream_)
{
	if (P < 0 || n_views < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || msg_stride < 0 || (msg_stride & 1)) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!means3D || !messages || !dL_dsh || (reinterpret_cast<uintptr_t>(messages) & 7)) return GSR_ERR_INVALID_ARG;
	PackedViews pk;
	pk.msgs = messages;
	pk.stride = msg_stride;
	return launch_sh_grad_from_views(P, D, M, n_views, means3D, nullptr, 0, nullptr, 0, scale, dL_dsh, nullptr, (hipStream_t)stream_,
	                                 nullptr, pk);
}

int gsr_sh_adam_from_packed_views(int P, int D, int M, int n_views, const float* means3D, const uint32_t* messages, long long msg_stride,
                                  float scale, float* shs, const gsr_sh_adam* o, void* stream_)
{
	if (P < 0 || n_views < 1 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || msg_stride < 0 || (msg_stride & 1)) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!means3D || !messages || (reinterpret_cast<uintptr_t>(messages) & 7) || !shs || !o || !o->exp_avg || !o->exp_avg_sq || o->step < 1)
		return GSR_ERR_INVALID_ARG;
	if (o->param && o->param != shs) return GSR_ERR_INVALID_ARG;
	const RowAdam ra = {shs, o->exp_avg, o->exp_avg_sq, adam_scalars(o->lr, o->lr_tail, o->beta1, o->beta2, o->eps, o->step)};
	LazyAdam la{};
	if (o->lazy) {
		gsr_sh_adam with_param = *o;
		with_param.param = shs;
		const int st = make_lazy_adam(with_param, shs, M, la);
		if (st != GSR_OK) return st;
	}
	PackedViews pk;
	pk.msgs = messages;
	pk.stride = msg_stride;
	return launch_sh_grad_from_views(P, D, M, n_views, means3D, nullptr, 0, nullptr, 0, scale, nullptr, &ra, (hipStream_t)stream_,
	                                 o->lazy ? &la : nullptr, pk);
}

int gsr_last_visible_count(void) { return t_last_visible; }
long long gsr_depth_resort_count(void) { return t_depth_resorts; }
int gsr_binning_tile_first(int raw_params, int P, int width, int height)
{
	if (P <= 0 || width <= 0 || height <= 0) return 0;
	return binning_tile_first(raw_params, P, div_up(width, TILE) * div_up(height, TILE)) ? 1 : 0;
}

int gsr_check_packed_views(int P, int n_views, const uint32_t* messages, long long msg_stride, int capacity_rows, void* stream_)
{
	if (P < 0 || n_views < 1 || msg_stride < PACK_HEADER || capacity_rows < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!messages) return GSR_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_;
	std::vector<uint32_t> head((size_t)n_views * PACK_HEADER);
	GSR_HIP(hipMemcpy2DAsync(head.data(), PACK_HEADER * sizeof(uint32_t), messages, (size_t)msg_stride * sizeof(uint32_t),
	                         PACK_HEADER * sizeof(uint32_t), (size_t)n_views, hipMemcpyDeviceToHost, stream));
	GSR_HIP(hipStreamSynchronize(stream));
	for (int v = 0; v < n_views; v++) {
		const uint32_t* h = &head[(size_t)v * PACK_HEADER];
		// K rows of a P-row view, all of them inside the `capacity_rows` rows that travelled: anything else was written for another
		// exchange, or the sender's capacity was too small and rows were dropped (h[3]).  (h[2] is the capacity the SENDER's buffer
		// had: gsr_backward writes its message before the ranks have agreed, into a buffer with room for every row.)
		if (h[1] != (uint32_t)P || h[3] != 0u || h[0] > h[2] || h[0] > (uint32_t)capacity_rows || h[0] > (uint32_t)P) return GSR_ERR_INVALID_ARG;
	}
	return GSR_OK;
}

int gsr_host_wait_stats(double* total_us, long long* calls, int reset)
{
	if (total_us) *total_us = t_sync_wait_us;
	if (calls) *calls = t_sync_waits;
	if (reset) {
		t_sync_wait_us = 0.0;
		t_sync_waits = 0;
	}
	return GSR_OK;
}

int gsr_sh_adam_lazy_slice(int P, const gsr_sh_adam* adam, int ahead, void* stream_)
{
	if (P < 0 || !adam || !adam->lazy) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	LazyAdam la{};
	int st = make_lazy_adam(*adam, nullptr, 16, la);
	if (st != GSR_OK) return st;
	return launch_sh_adam_lazy(P, nullptr, la, (hipStream_t)stream_, ahead ? 3 : 2);
}

int gsr_profile_enable(int on)
{
	if (on) {
		int st = t_prof.create();
		if (st != GSR_OK) return st;
	}
	t_prof.on = on == 2 ? 2 : (on != 0 ? 1 : 0);
	t_prof.fwd_done = t_prof.bwd_done = false;
	return GSR_OK;
}
int gsr_profile_stage_count(void) { return ST_COUNT; }
const char* gsr_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? k_stage_names[i] : ""; }
int gsr_profile_read(float* ms, int count)
{
	if (!ms || count < ST_COUNT) return GSR_ERR_INVALID_ARG;
	for (int i = 0; i < ST_COUNT; i++) ms[i] = -1.f;
	if (t_prof.fwd_done) {
		GSR_HIP(hipEventSynchronize(t_prof.fwd[ST_FWD_COUNT]));
		for (int i = 0; i < ST_FWD_COUNT; i++) GSR_HIP(hipEventElapsedTime(&ms[i], t_prof.fwd[i], t_prof.fwd[i + 1]));
	}
	if (t_prof.bwd_done && t_prof.on == 2) {
		GSR_HIP(hipEventSynchronize(t_prof.bwd[2]));
		GSR_HIP(hipEventElapsedTime(&ms[ST_BLEND_BWD], t_prof.bwd[1], t_prof.bwd[2]));
	} else if (t_prof.bwd_done) {
		GSR_HIP(hipEventSynchronize(t_prof.bwd[3]));
		for (int i = 0; i < 3; i++) GSR_HIP(hipEventElapsedTime(&ms[ST_FWD_COUNT + i], t_prof.bwd[i], t_prof.bwd[i + 1]));
	}
	return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
	(void)projmatrix;  // the reference projects and discards (auxiliary.h:148-151); only view-space z decides
	if (P < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!means3D || !viewmatrix || !present) return GSR_ERR_INVALID_ARG;
	return launch_check_frustum(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int gsr_knn_mean_dist2(int P, const float* points, float* meanDists, gsr_alloc_fn scratchBuffer, void* scratch_ctx,
                       void* stream)
{
	if (P < 0) return GSR_ERR_INVALID_ARG;
	if (P == 0) return GSR_OK;
	if (!points || !meanDists || !scratchBuffer) return GSR_ERR_INVALID_ARG;
	char* scratch = scratchBuffer(scratch_ctx, knn_scratch_bytes(P));
	if (!scratch) return GSR_ERR_ALLOC;
	return launch_knn(P, points, meanDists, scratch, (hipStream_t)stream);
}

}  // extern "C"
