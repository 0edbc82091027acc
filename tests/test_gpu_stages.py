"""Scan / radix sort / kNN / markVisible of the HIP build at sizes the pipeline tests do not reach."""
import numpy as np
import pytest
import torch

import stages
from photo_slam_amd import capi, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    capi.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [1, 255, 2049, 1_000_003, 5_000_000])
def test_scan(dev, n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 50, n).astype(np.uint32)
    for inclusive in (0, 1):
        got = stages.scan_u32(None, dev, v, inclusive)
        want = np.cumsum(v, dtype=np.uint64).astype(np.uint32) - (0 if inclusive else v)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("n,begin,end", [(1, 0, 8), (4097, 0, 13), (100_000, 3, 9), (2_000_000, 0, 32), (8_000_001, 0, 13)])
def test_radix_sort_stable(dev, n, begin, end):
    rng = np.random.default_rng(n + end)
    keys = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if n > 50:
        keys[rng.integers(0, n, n // 2)] = keys[0]
    vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    k, v = stages.radix_sort_pairs(None, dev, keys, vals, begin, end)
    wk, wv = stages.reference_sort(keys, vals, begin, end)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)
    k, v = stages.radix_sort_pairs(None, dev, keys, None, begin, end)
    wk, wv = stages.reference_sort(keys, None, begin, end)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)


def test_radix_sort_skewed_digits(dev):
    n = 3_000_000
    keys = (np.uint32(0x3F800000) + (np.arange(n) % 7).astype(np.uint32)).astype(np.uint32)
    k, v = stages.radix_sort_pairs(None, dev, keys, None, 0, 32)
    wk, wv = stages.reference_sort(keys, None, 0, 32)
    assert np.array_equal(k, wk) and np.array_equal(v, wv)


@pytest.mark.parametrize("P", [1, 3, 4, 1025, 100_000])
def test_knn_matches_oracle(oracle, dev, P):
    rng = np.random.default_rng(P)
    pts = (rng.standard_normal((P, 3)) * [3, 1, 2] + [0.5, -0.2, 1.0]).astype(np.float32)
    if P > 10:
        pts[5] = pts[6]
    got = stages.knn(None, dev, pts)
    want = oracle.knn(pts)
    assert np.array_equal(got, want)   # distances are IEEE ops in the same order on both sides: bit-exact


def test_mark_visible(oracle, dev):
    cl = scene.make_cloud(300_000, 64, 48, 50.0, 50.0, seed=5)
    cam = cl.cameras[0]
    got = stages.mark_visible(None, dev, cl.xyz, cam.viewmatrix, cam.projmatrix)
    assert np.array_equal(got, oracle.mark_visible(cl.xyz, cam.viewmatrix, cam.projmatrix))


def test_tile_depth_sort_at_its_path_boundaries(dev):
    from test_emu_stages import tile_depth_sort_case
    tile_depth_sort_case(None, dev, seed=1, big=300_000)
