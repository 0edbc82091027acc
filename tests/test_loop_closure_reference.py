"""GaussianModel::applyScaledTransformation / scaledTransformVisiblePointsOfKeyframe (src/gaussian_model.cpp:379-475, the
loop-closure correction of the map) in both hosts against the REFERENCE's own member functions -- extracted verbatim from
src/gaussian_model.cpp and compiled into oracle/_ref/libref_host_{emu,hip}.so (oracle/build_ref.py: build_host_tree;
applyScaledTransformation against the stand-in Sophus::SE3f of oracle/ref_host/sophus_standin.h).  Both sides call the point
kernels through the same boundary functions (transformPoints / scaleAndTransformThenMarkVisiblePoints of lib/libcuda_rasterizer,
bit-pinned to the reference's kernels by tests/test_points_and_ply.py), so what is compared here is the host logic: which
points move, the flags, the quirks (log-scales MULTIPLIED by s, rotation_ replaced by the normalised quaternions) and the
surgery on the Adam state (moments of the replaced groups zeroed, step counters kept, the other groups untouched).

Bars: every tensor bit-equal."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd import scene
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref_ops(kind):
    from oracle import build_ref
    path = build_ref.build_host_tree().get(kind)
    if path is None or not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_host_*.so was never built (no reference tree, no prebuilt library)")
    torch.ops.load_library(path)
    return getattr(torch.ops, build_ref.HOST_OPS[kind])


def _host(variant):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    return load_host(variant)


def _state(cl, dev, seed):
    """a model in the middle of training: parameters, non-trivial Adam moments and step counters, ages"""
    g = torch.Generator().manual_seed(seed)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    params6 = [t(cl.xyz), t(cl.features_dc), t(cl.features_rest), t(cl.opacity), t(cl.scaling), t(cl.rotation)]
    m6 = [(0.01 * torch.randn(p.shape, generator=g)).to(dev) for p in params6]
    v6 = [(1e-4 * torch.rand(p.shape, generator=g)).to(dev) for p in params6]
    steps6 = [7, 5, 5, 6, 7, 4]
    exist = torch.randint(0, 40, (cl.xyz.shape[0],), generator=g, dtype=torch.int32).to(dev)
    return params6, m6, v6, steps6, exist


def _rigid(seed):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R.astype(np.float32)
    T[:3, 3] = rng.standard_normal(3).astype(np.float32) * 0.3
    return T


def _ref_session(rops, cl, dev, state):
    params6, m6, v6, steps6, exist = state
    h = rops.create(params6, 3, 3, float(cl.extent), float(cl.extent))
    rops.training_setup(h, {})
    rops.set_adam_state(h, m6, v6, steps6)
    rops.set_exist_since_iter(h, exist)
    return h


def _five(six):
    """the reference's six groups as this repository's five (features_dc and features_rest are one [P,16,3] leaf)"""
    return [six[0], torch.cat([six[1], six[2]], 1), six[3], six[4], six[5]]


def _cpp_session(ops, cl, dev, state):
    params6, m6, v6, steps6, exist = state
    p5 = _five(params6)
    h = ops.trainer_create(p5[0], p5[1], p5[2], p5[3], p5[4], 3, float(cl.extent), torch.zeros(3, device=dev))
    mom = ops.trainer_moments(h)
    for dst, src in zip(mom, _five(m6) + _five(v6)):
        dst.copy_(src)
    ops.trainer_set_steps(h, [steps6[0], steps6[1], steps6[3], steps6[4], steps6[5]])
    ops.trainer_set_exist_since_iter(h, exist)
    return h


def _py_model(cl, dev, state):
    params6, m6, v6, steps6, exist = state
    g = GaussianModel.from_cloud(cl, device=dev)
    g.trainingSetup(GaussianOptimizationParams())
    leaves = [g.xyz_, g.features_, g.opacity_, g.scaling_, g.rotation_]
    for p, m, v, s in zip(leaves, _five(m6), _five(v6), [steps6[0], steps6[1], steps6[3], steps6[4], steps6[5]]):
        g.optimizer_.state[id(p)] = dict(exp_avg=m.clone(), exp_avg_sq=v.clone(), step=s)
    g.exist_since_iter_ = exist.clone()
    return g


def _compare(name, want_dump, params5, moments10, steps5):
    w_params, w_m, w_v, w_steps = want_dump[0:6], want_dump[6:12], want_dump[12:18], want_dump[22].tolist()
    for k, (w, got) in enumerate(zip(_five(w_params), params5)):
        assert torch.equal(w.detach().cpu(), got.detach().cpu()), (name, "parameter", k)
    for k, (w, got) in enumerate(zip(_five(w_m) + _five(w_v), moments10)):
        assert torch.equal(w.cpu(), got.cpu()), (name, "moment", k)
    assert [w_steps[0], w_steps[1], w_steps[3], w_steps[4], w_steps[5]] == list(steps5), (name, w_steps, steps5)
    assert w_steps[1] == w_steps[2]


def run(dev, lib_path, kind, variant, cl):
    rops, ops = _ref_ops(kind), _host(variant)
    state = _state(cl, dev, 5)
    T = _rigid(1)
    cam = cl.cameras[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    view, proj = t(cam.viewmatrix), t(cam.projmatrix)
    diff_pose = t(np.ascontiguousarray(_rigid(2).T))      # the transposed matrix, as the mapper hands it over
    flags0 = (torch.rand(cl.xyz.shape[0], generator=torch.Generator().manual_seed(3)) < 0.8).to(dev)
    rp._LIB_OVERRIDE = lib_path
    try:
        # ---- the reference's own functions
        hr = _ref_session(rops, cl, dev, state)
        rops.apply_scaled_transformation(hr, 1.25, t(T))
        after_scale = [x.clone() for x in rops.dump(hr)]   # (the next call works on the leaf's storage in place, :430-447)
        flags_ref, moved_ref = rops.scaled_transform_visible(hr, flags0, diff_pose, view, proj, 20, 15, 1.1)
        after_move = [x.clone() for x in rops.dump(hr)]
        rops.destroy(hr)
        assert moved_ref > 0 and bool((flags_ref != flags0).any()), "nothing moved: the case would not test the call"
        # ---- the C++ host
        hc = _cpp_session(ops, cl, dev, state)
        ops.trainer_apply_scaled_transformation(hc, 1.25, t(T))
        _compare("c++ applyScaledTransformation", after_scale, ops.trainer_params(hc), ops.trainer_moments(hc), ops.trainer_steps(hc))
        flags_c, moved_c = ops.trainer_scaled_transform_visible(hc, flags0, diff_pose, view, proj, 20, 15, 1.1)
        assert moved_c == moved_ref and torch.equal(flags_c, flags_ref)
        _compare("c++ scaledTransformVisiblePointsOfKeyframe", after_move, ops.trainer_params(hc), ops.trainer_moments(hc),
                 ops.trainer_steps(hc))
        ops.trainer_destroy(hc)
        # ---- the Python host
        g = _py_model(cl, dev, state)
        leaves = lambda: [g.xyz_, g.features_, g.opacity_, g.scaling_, g.rotation_]
        mom = lambda: [g.optimizer_.state[id(p)]["exp_avg"] for p in leaves()] + [g.optimizer_.state[id(p)]["exp_avg_sq"] for p in leaves()]
        stp = lambda: [int(g.optimizer_.state[id(p)]["step"]) for p in leaves()]
        g.applyScaledTransformation(1.25, t(T))
        _compare("python applyScaledTransformation", after_scale, leaves(), mom(), stp())
        flags_p = flags0.clone()
        moved_p = g.scaledTransformVisiblePointsOfKeyframe(flags_p, diff_pose, view, proj, 20, 15, 0, 1.1)
        assert moved_p == moved_ref and torch.equal(flags_p, flags_ref)
        _compare("python scaledTransformVisiblePointsOfKeyframe", after_move, leaves(), mom(), stp())
    finally:
        rp._LIB_OVERRIDE = None
    print(f"[loop closure, {kind}] {moved_ref} of {cl.xyz.shape[0]} points moved by the keyframe correction")


def run_in_arena(dev, lib_path, variant, cl):
    """The same calls + resetOpacity on leaves that LIVE IN THE ARENA (after an increasePcd): the values are replaced in place, the
    results equal those of a model whose leaves were moved out of the arena first (release: the path the reference comparison
    above pins), and the next increasePcd still appends in place -- it used to rebuild the whole arena (+25 ms at 4 M Gaussians)."""
    ops = _host(variant)
    state = _state(cl, dev, 5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cam = cl.cameras[0]
    view, proj = t(cam.viewmatrix), t(cam.projmatrix)
    T, diff_pose = t(_rigid(1)), t(np.ascontiguousarray(_rigid(2).T))
    gen = torch.Generator().manual_seed(11)
    new_pts = (torch.rand(40, 3, generator=gen) * 2.0 - 1.0).to(dev)
    new_cols = torch.rand(40, 3, generator=gen).to(dev)
    more_pts, more_cols = new_pts[:7] + 0.01, new_cols[:7]
    flags0 = (torch.rand(cl.xyz.shape[0] + 40, generator=torch.Generator().manual_seed(3)) < 0.8).to(dev)
    ptrs = lambda ts: [x.data_ptr() for x in ts]
    rp._LIB_OVERRIDE = lib_path
    try:
        def cpp(release):
            h = _cpp_session(ops, cl, dev, state)
            ops.trainer_increase_pcd(h, new_pts, new_cols, 12, False)      # the leaves move into the arena
            if release:
                ops.trainer_release_arena(h)
            before = ptrs(ops.trainer_params(h)) + ptrs(ops.trainer_moments(h))
            ops.trainer_apply_scaled_transformation(h, 1.25, T)
            flags, moved = ops.trainer_scaled_transform_visible(h, flags0, diff_pose, view, proj, 20, 15, 1.1)
            ops.trainer_reset_opacity(h)
            after = ptrs(ops.trainer_params(h)) + ptrs(ops.trainer_moments(h))
            out = [x.detach().clone() for x in list(ops.trainer_params(h)) + list(ops.trainer_moments(h))] + [flags.clone()]
            if not release:
                assert after == before, "a leaf or a moment left the arena"
                ops.trainer_increase_pcd(h, more_pts, more_cols, 13, False)
                assert ptrs(ops.trainer_params(h)) + ptrs(ops.trainer_moments(h)) == before, "the append behind the calls rebuilt the arena"
            ops.trainer_destroy(h)
            return out, moved

        def py(release):
            g = _py_model(cl, dev, state)
            leaves = lambda: [g.xyz_, g.features_, g.opacity_, g.scaling_, g.rotation_]
            mom = lambda: [g.optimizer_.state[id(p)]["exp_avg"] for p in leaves()] + [g.optimizer_.state[id(p)]["exp_avg_sq"] for p in leaves()]
            g.increasePcd(new_pts, new_cols, 12)
            if release:
                g.release_arena()
            before = ptrs(leaves()) + ptrs(mom())
            g.applyScaledTransformation(1.25, T)
            flags = flags0.clone()
            moved = g.scaledTransformVisiblePointsOfKeyframe(flags, diff_pose, view, proj, 20, 15, 0, 1.1)
            g.resetOpacity()
            out = [x.detach().clone() for x in leaves() + mom()] + [flags]
            if not release:
                assert ptrs(leaves()) + ptrs(mom()) == before, "a leaf or a moment left the arena"
                g.increasePcd(more_pts, more_cols, 13)
                assert ptrs(leaves()) + ptrs(mom()) == before, "the append behind the calls rebuilt the arena"
            return out, moved

        for name, host in (("c++", cpp), ("python", py)):
            (a, moved_a), (b, moved_b) = host(False), host(True)
            assert moved_a == moved_b and moved_a > 0
            for k, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x.cpu(), y.cpu()), (name, k)
            # the replaced groups' moments are zero, the others' are not (xyz, opacity, scaling, rotation: 0, 2, 3, 4)
            for k in (0, 2, 3, 4):
                assert not a[5 + k].any() and not a[10 + k].any(), (name, k)
            assert a[5 + 1][: cl.xyz.shape[0]].any()
    finally:
        rp._LIB_OVERRIDE = None


def test_loop_closure_and_reset_keep_the_leaves_in_the_arena_on_the_emulator(emu_lib_path):
    cl = scene.make_cloud(500, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35)
    run_in_arena(torch.device("cpu"), emu_lib_path, "emu", cl)


@pytest.mark.gpu
def test_loop_closure_and_reset_keep_the_leaves_in_the_arena_on_gpu():
    cl = scene.make_config("C1", seed=0)
    run_in_arena(torch.device("cuda:0"), None, "hip", cl)


def test_loop_closure_methods_match_the_reference_on_the_emulator(emu_lib_path):
    cl = scene.make_cloud(500, 48, 32, 40.0, 40.0, seed=3, scale_k=0.35)
    run(torch.device("cpu"), emu_lib_path, "emu", "emu", cl)


@pytest.mark.gpu
def test_loop_closure_methods_match_the_reference_on_gpu():
    cl = scene.make_config("C1", seed=0)
    run(torch.device("cuda:0"), None, "hip", "hip", cl)
