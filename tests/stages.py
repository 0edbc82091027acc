"""Helpers to call the TEST-ONLY stage-level entry points (tests/dev/gsr_dev.h) and the kNN /
markVisible wrappers on a given build (HIP library on the GPU, emulator on the host)."""
import ctypes as C

import numpy as np
import torch

import os
import sys

from photo_slam_amd import capi
from photo_slam_amd import rasterize_points as rp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev"))
import devapi  # noqa: E402


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)


def scan_u32(lib_path, dev, values, inclusive):
    lib = devapi.load(lib_path)
    n = int(values.shape[0])
    src = torch.from_numpy(values.astype(np.uint32).view(np.int32)).to(dev)
    out = torch.zeros_like(src)
    scratch = torch.empty(int(lib.gsr_scan_scratch_bytes(n)) + 16, dtype=torch.uint8, device=dev)
    st = lib.gsr_stage_scan_u32(src.data_ptr(), out.data_ptr(), n, int(inclusive), scratch.data_ptr(), _stream(dev))
    capi.check(capi.load(lib_path), st, "gsr_stage_scan_u32")
    return out.cpu().numpy().view(np.uint32)


def radix_sort_pairs(lib_path, dev, keys, values, begin_bit, end_bit):
    lib = devapi.load(lib_path)
    n = int(keys.shape[0])
    k_in = torch.from_numpy(keys.astype(np.uint32).view(np.int32)).to(dev)
    v_in = None if values is None else torch.from_numpy(values.astype(np.uint32).view(np.int32)).to(dev)
    k_out = torch.zeros_like(k_in)
    v_out = torch.zeros_like(k_in)
    scratch = torch.empty(int(lib.gsr_sort_scratch_bytes(n)) + 16, dtype=torch.uint8, device=dev)
    st = lib.gsr_stage_radix_sort_pairs(k_in.data_ptr(), None if v_in is None else v_in.data_ptr(), k_out.data_ptr(),
                                        v_out.data_ptr(), n, begin_bit, end_bit, scratch.data_ptr(), _stream(dev))
    capi.check(capi.load(lib_path), st, "gsr_stage_radix_sort_pairs")
    assert np.array_equal(k_in.cpu().numpy().view(np.uint32), keys.astype(np.uint32)), "keys_in was modified"
    return k_out.cpu().numpy().view(np.uint32), v_out.cpu().numpy().view(np.uint32)


def tile_depth_sort(lib_path, dev, lengths, depth_key, point_list):
    """the per-tile depth sort (tile_depth_sort.hip) on lists of the given lengths laid end to end in point_list"""
    lib = devapi.load(lib_path)
    ends = np.cumsum(np.asarray(lengths, np.int64))
    ranges = np.stack([ends - np.asarray(lengths, np.int64), ends], 1).astype(np.uint32)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.uint32).view(np.int32)).to(dev)
    r, k, pl = to(ranges), to(depth_key), to(point_list)
    n = int(point_list.shape[0])
    spare = [torch.full((n + 64,), -1, dtype=torch.int32, device=dev) for _ in range(3)]
    st = lib.gsr_stage_tile_depth_sort(r.data_ptr(), int(len(lengths)), k.data_ptr(), pl.data_ptr(), spare[0].data_ptr(), spare[1].data_ptr(),
                                       spare[2].data_ptr(), _stream(dev))
    capi.check(capi.load(lib_path), st, "gsr_stage_tile_depth_sort")
    return pl.cpu().numpy().view(np.uint32)


def reference_sort(keys, values, begin_bit, end_bit):
    """what a stable sort on the bit field [begin_bit, end_bit) must produce"""
    mask = np.uint64((1 << (end_bit - begin_bit)) - 1)
    field = (keys.astype(np.uint64) >> np.uint64(begin_bit)) & mask
    perm = np.argsort(field, kind="stable")
    vals = np.arange(keys.shape[0], dtype=np.uint32) if values is None else values
    return keys[perm], vals[perm]


def knn(lib_path, dev, points):
    rp._LIB_OVERRIDE = lib_path
    try:
        return rp.distCUDA2(torch.from_numpy(np.ascontiguousarray(points, np.float32)).to(dev)).cpu().numpy()
    finally:
        rp._LIB_OVERRIDE = None


def mark_visible(lib_path, dev, means3D, view, proj):
    rp._LIB_OVERRIDE = lib_path
    try:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        return rp.markVisible(t(means3D), t(view), t(proj)).cpu().numpy()
    finally:
        rp._LIB_OVERRIDE = None
