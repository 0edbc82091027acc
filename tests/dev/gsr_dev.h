/*
 * gsr_dev.h -- TEST-ONLY introspection of the rasterizer (tests/dev/libgsr_dev.so, and part of the emulator build): typed
 * views into the opaque scratch buffers of gsr_forward, so that the parity tests can compare every intermediate of the
 * pipeline with the oracle (tile rectangles, sort order, tile ranges, n_contrib, ...), and the scan / radix-sort building
 * blocks at sizes the full pipeline does not reach in a unit test.  NOT part of the product: libgsr_hip.so exports none of
 * these; libgsr_dev.so is a thin host-side library that links against it.  The scratch layout is an implementation detail
 * and may change with any commit.  Same conventions as gsr.h (device pointers, explicit stream, int status).
 */
#ifndef GSR_DEV_H
#define GSR_DEV_H
#include "../../include/gsr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Typed views into the opaque scratch buffers of gsr_forward.  The layout is this
 * library's own (it replaces GeometryState / BinningState / ImageState,
 * cuda_rasterizer/rasterizer_impl.h:32-62); the reference's contract is only
 * "same bytes back in backward". */
typedef struct gsr_geometry_view {
	uint32_t* depth_key;     /* [P]  float bits of view-space z; 0xFFFFFFFF if culled           */
	uint32_t* tiles_touched; /* [P]                                                             */
	int*      radii;         /* [P]  internal copy of radii                                     */
	uint16_t* rect;          /* [P][4] tile rect (min.x, min.y, max.x, max.y), getRect output   */
	float*    rec;           /* [P][12] blend record: x,y,conic.x,conic.y | conic.z,opacity,r,g | b,-,-,- */
	float*    cov3D;         /* [P][6]                                                          */
	uint8_t*  clamped;       /* [P]  bit c set = colour channel c was clamped                   */
	uint32_t* order;         /* [P]  Gaussian ids in (depth, id) order                          */
	uint32_t* offsets;       /* [P]  exclusive scan of tiles_touched[order[.]]                  */
} gsr_geometry_view;
typedef struct gsr_binning_view {
	uint32_t* point_list;    /* [R] Gaussian id per instance, sorted by (tile, depth, id)       */
	uint32_t* tile_keys;     /* [R] tile id per sorted instance                                 */
} gsr_binning_view;
typedef struct gsr_image_view {
	float*    final_T;       /* [H*W] */
	uint32_t* n_contrib;     /* [H*W] */
	uint32_t* ranges;        /* [T][2] */
} gsr_image_view;

int gsr_view_geometry(char* geom_buffer, int P, gsr_geometry_view* out);
int gsr_view_binning(char* binning_buffer, int R, int width, int height, gsr_binning_view* out);
int gsr_view_image(char* image_buffer, int width, int height, gsr_image_view* out);

/* Exclusive / inclusive prefix sum of n uint32 (the cub::DeviceScan::InclusiveSum
 * replacement, rasterizer_impl.cu:276).  scratch: gsr_scan_scratch_bytes(n). */
size_t gsr_scan_scratch_bytes(int n);
int gsr_stage_scan_u32(const uint32_t* in, uint32_t* out, int n, int inclusive, char* scratch, void* stream);

/* Stable LSD radix sort of (u32 key, u32 value) pairs on key bits [begin_bit,end_bit)
 * (the cub::DeviceRadixSort::SortPairs replacement, rasterizer_impl.cu:303-308 and
 * simple_knn.cu:210-213).  values_in == NULL means values = 0..n-1.  Inputs are only
 * read and must not alias the outputs (GSR_ERR_INVALID_ARG: with an odd number of passes the first pass scatters straight
 * into the output pair); the result is in keys_out/values_out.  scratch: gsr_sort_scratch_bytes(n). */
size_t gsr_sort_scratch_bytes(int n);
int gsr_stage_radix_sort_pairs(const uint32_t* keys_in, const uint32_t* values_in, uint32_t* keys_out, uint32_t* values_out,
                               int n, int begin_bit, int end_bit, char* scratch, void* stream);

/* The per-tile depth sort of the tile-first binning arrangement (csrc/tile_depth_sort.hip): list t = point_list[ranges[t].x ..
 * ranges[t].y) is sorted in place by depth_key[id], stable (equal keys keep their order).  spare_keys / spare_values /
 * spare_words: three arrays of as many words as point_list (scratch of the lists beyond 2 048 entries). */
int gsr_stage_tile_depth_sort(const uint32_t* ranges /* [tiles][2] */, int tiles, const uint32_t* depth_key, uint32_t* point_list,
                              uint32_t* spare_keys, uint32_t* spare_values, uint32_t* spare_words, void* stream);

#ifdef __cplusplus
}
#endif
#endif
