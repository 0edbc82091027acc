// gsr_dev.cpp -- TEST-ONLY introspection and building-block entry points (tests/dev/gsr_dev.h).  Host code only: the
// views carve the caller's scratch buffers with the product's own layout functions (csrc/state.h), the scan / sort hooks
// call the product's launchers (exported C++ symbols of libgsr_hip.so, which this library links against).  The emulator
// build (tests/emu) compiles this file into libgsr_emu.so.
#include "state.h"
#include "gsr_dev.h"

using namespace gsr;

extern "C" {

size_t gsr_scan_scratch_bytes(int n) { return scan_scratch_elems(n < 0 ? 0 : n) * sizeof(uint32_t); }
size_t gsr_sort_scratch_bytes(int n)
{
	const size_t m = (size_t)(n < 0 ? 0 : n);
	return (sort_scratch_elems((int)m) + 2 * m + 64) * sizeof(uint32_t);  // histograms + one temp (key, value) buffer pair
}

int gsr_view_geometry(char* geom_buffer, int P, gsr_geometry_view* out)
{
	if (!geom_buffer || P < 0 || !out) return GSR_ERR_INVALID_ARG;
	GeometryState g = GeometryState::carve(geom_buffer, (size_t)P);
	out->depth_key = g.depth_key; out->tiles_touched = g.tiles_touched; out->radii = g.radii; out->rect = g.rect;
	out->rec = reinterpret_cast<float*>(g.rec); out->cov3D = g.cov3D; out->clamped = g.clamped; out->order = g.order;
	out->offsets = g.offsets;
	return GSR_OK;
}
int gsr_view_binning(char* binning_buffer, int R, int width, int height, gsr_binning_view* out)
{
	if (!binning_buffer || R < 0 || !out || width <= 0 || height <= 0) return GSR_ERR_INVALID_ARG;
	BinningState b = BinningState::carve(binning_buffer, (size_t)R);
	const int passes = tile_sort_passes(div_up(width, TILE) * div_up(height, TILE), R);
	out->point_list = (passes % 2) ? b.vals_b : b.vals_a;
	out->tile_keys = (passes % 2) ? b.keys_b : b.keys_a;
	return GSR_OK;
}
int gsr_view_image(char* image_buffer, int width, int height, gsr_image_view* out)
{
	if (!image_buffer || !out || width <= 0 || height <= 0) return GSR_ERR_INVALID_ARG;
	const size_t T = (size_t)div_up(width, TILE) * div_up(height, TILE);
	ImageState im = ImageState::carve(image_buffer, (size_t)width * height, T);
	out->final_T = im.final_T; out->n_contrib = im.n_contrib; out->ranges = reinterpret_cast<uint32_t*>(im.ranges);
	return GSR_OK;
}
int gsr_stage_scan_u32(const uint32_t* in, uint32_t* out, int n, int inclusive, char* scratch, void* stream)
{
	if (n < 0 || (n > 0 && (!in || !out || !scratch))) return GSR_ERR_INVALID_ARG;
	return launch_scan_u32(in, nullptr, out, n, inclusive != 0, reinterpret_cast<uint32_t*>(scratch), (hipStream_t)stream);
}
int gsr_stage_radix_sort_pairs(const uint32_t* keys_in, const uint32_t* values_in, uint32_t* keys_out, uint32_t* values_out, int n,
                               int begin_bit, int end_bit, char* scratch, void* stream_)
{
	if (n < 0 || begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit) return GSR_ERR_INVALID_ARG;
	if (n == 0) return GSR_OK;
	if (!keys_in || !keys_out || !values_out || !scratch) return GSR_ERR_INVALID_ARG;
	if (keys_in == keys_out || (values_in && values_in == values_out)) return GSR_ERR_INVALID_ARG;   // in place is not supported
	hipStream_t stream = (hipStream_t)stream_;
	const int passes = div_up(end_bit - begin_bit, RADIX_BITS);
	// keys_in/values_in are only read; a temp pair carved from scratch is the other half of the
	// ping-pong, arranged so that the final pass lands in (keys_out, values_out).
	uint32_t* sc = reinterpret_cast<uint32_t*>(scratch);
	uint32_t* tmp_k = sc + sort_scratch_elems(n);
	uint32_t* tmp_v = tmp_k + n;
	uint32_t *kp, *vp, *kq, *vq;  // ping, pong
	if (passes % 2) { kq = keys_out; vq = values_out; kp = tmp_k; vp = tmp_v; }
	else            { kp = keys_out; vp = values_out; kq = tmp_k; vq = tmp_v; }
	uint32_t *kres, *vres;
	return launch_radix_sort(keys_in, values_in, kp, vp, kq, vq, n, begin_bit, end_bit, sc, stream, &kres, &vres);
}

int gsr_stage_tile_depth_sort(const uint32_t* ranges, int tiles, const uint32_t* depth_key, uint32_t* point_list, uint32_t* spare_keys,
                              uint32_t* spare_values, uint32_t* spare_words, void* stream)
{
	return launch_tile_depth_sort(reinterpret_cast<const uint2*>(ranges), tiles, depth_key, point_list, spare_keys, spare_values, spare_words,
	                              (hipStream_t)stream);
}

}  // extern "C"
