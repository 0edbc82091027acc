"""Densification / prune / opacity reset / Adam pinned to the REFERENCE's own code.

oracle/_ref/libref_densify{,_cuda}.so holds GaussianModel::{densifyAndPrune, densifyAndClone, densifyAndSplit,
densificationPostfix, prunePoints, resetOpacity, replaceTensorToOptimizer, addDensificationStats, trainingSetup} extracted
verbatim from /root/reference/src/gaussian_model.cpp and compiled against LibTorch (oracle/ref_densify.cpp,
oracle/build_ref.py:build_densify) -- the three-rebuild implementation with its Adam-state surgery and torch::optim::Adam.
BOTH hosts of this repository (the Python mirror photo-slam_amd/gaussian_model.py and the C++ host
photo-slam_amd/host/src/gaussian_model_densify.cpp, each one rebuild per call on the HIP stream-compaction kernels) must
produce the same tensors, the same moments and the same step counters from the same state and the same at::normal seed.

Bars: parameters that are copied (everything except the split children's positions / scales) bit-equal; the children's
positions to 1e-6 (the reference multiplies R(q) * sample with torch::bmm, the hosts with explicit FMAs), their scales to
1e-6 relative (log(s / 1.6) both sides)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("xyz_", "features_", "opacity_", "scaling_", "rotation_")


def _ref_ops(kind):
    from oracle import ref_model
    ops = ref_model.load(kind)
    if ops is None:
        pytest.skip("oracle/_ref/libref_densify*.so was never built (no reference tree, no prebuilt library)")
    return ref_model, ops


def make_state(ref_model, P, seed, dev, scale_split=0.5):
    """A trained-looking model: random parameters, non-trivial Adam moments, statistics accumulated over a few views."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    params = [r(P, 3), 0.5 * r(P, 1, 3), 0.05 * r(P, 15, 3), 2.0 * r(P, 1), -2.0 + scale_split * r(P, 3), r(P, 4)]
    m = [0.01 * r(*p.shape) for p in params]
    v = [1e-4 * torch.rand(*p.shape, generator=g) for p in params]
    denom = torch.randint(0, 5, (P, 1), generator=g).float()          # zeros included: 0/0 -> nan -> 0 (:801-802)
    accum = torch.rand(P, 1, generator=g) * denom * 4e-4
    accum[denom == 0] = 0
    max_radii = torch.rand(P, generator=g) * 40
    st = ref_model.State([t.to(dev) for t in params], [t.to(dev) for t in m], [t.to(dev) for t in v], [7] * 6, accum.to(dev),
                         denom.to(dev), max_radii.to(dev), torch.randint(0, 900, (P,), generator=g, dtype=torch.int32).to(dev))
    return st


def python_host(st, dev):
    g = GaussianModel(3, device=dev)
    leaf = lambda t: t.detach().clone().contiguous().requires_grad_(True)
    g.xyz_, g.features_, g.opacity_ = leaf(st.params[0]), leaf(st.features), leaf(st.params[3])
    g.scaling_, g.rotation_ = leaf(st.params[4]), leaf(st.params[5])
    g.active_sh_degree_ = 3
    g.xyz_gradient_accum_, g.denom_, g.max_radii2D_ = st.accum.clone(), st.denom.clone(), st.max_radii2D.clone()
    g.exist_since_iter_ = st.exist_since_iter.clone()
    g.trainingSetup(GaussianOptimizationParams())
    cat = lambda l: torch.cat([l[1], l[2]], 1)
    for name, m, v in zip(NAMES, (st.exp_avg[0], cat(st.exp_avg), st.exp_avg[3], st.exp_avg[4], st.exp_avg[5]),
                          (st.exp_avg_sq[0], cat(st.exp_avg_sq), st.exp_avg_sq[3], st.exp_avg_sq[4], st.exp_avg_sq[5])):
        g.optimizer_.state[id(getattr(g, name))] = dict(exp_avg=m.clone(), exp_avg_sq=v.clone(), step=st.steps[0])
    return g


def python_state(g):
    params = [getattr(g, n).detach() for n in NAMES]
    mom = [g.optimizer_.moments(getattr(g, n)) for n in NAMES]
    steps = [g.optimizer_.state[id(getattr(g, n))]["step"] for n in NAMES]
    return params, [m for m, _ in mom], [v for _, v in mom], steps, (g.xyz_gradient_accum_, g.denom_, g.max_radii2D_), g.exist_since_iter_


def cpp_host(ops, st, dev):
    h = ops.trainer_create(st.params[0], st.features.contiguous(), st.params[3], st.params[4], st.params[5], 3, 1.0,
                           torch.zeros(3, device=dev))
    for dst, src in zip(ops.trainer_stats(h), (st.accum, st.denom, st.max_radii2D)):
        dst.copy_(src)
    mom = ops.trainer_moments(h)
    cat = lambda l: torch.cat([l[1], l[2]], 1)
    for i, (m, v) in enumerate(zip((st.exp_avg[0], cat(st.exp_avg), st.exp_avg[3], st.exp_avg[4], st.exp_avg[5]),
                                   (st.exp_avg_sq[0], cat(st.exp_avg_sq), st.exp_avg_sq[3], st.exp_avg_sq[4], st.exp_avg_sq[5]))):
        mom[i].copy_(m)
        mom[5 + i].copy_(v)
    ops.trainer_set_steps(h, [st.steps[0]] * 5)
    ops.trainer_set_exist_since_iter(h, st.exist_since_iter)
    return h


def cpp_state(ops, h):
    mom = ops.trainer_moments(h)
    return (list(ops.trainer_params(h)), list(mom[:5]), list(mom[5:]), list(ops.trainer_steps(h)), tuple(ops.trainer_stats(h)),
            ops.trainer_exist_since_iter(h))


def compare(ref, got, n_children, what):
    """ref: ref_model.State after the reference's call; got: (params[5], m[5], v[5], steps[5], stats) of one of our hosts."""
    params, m, v, steps, stats, exist = got
    assert exist.dtype == torch.int32 and torch.equal(exist, ref.exist_since_iter), (what, "exist_since_iter_")
    cat = lambda l: torch.cat([l[1], l[2]], 1)
    want_p = (ref.params[0], cat(ref.params), ref.params[3], ref.params[4], ref.params[5])
    want_m = (ref.exp_avg[0], cat(ref.exp_avg), ref.exp_avg[3], ref.exp_avg[4], ref.exp_avg[5])
    want_v = (ref.exp_avg_sq[0], cat(ref.exp_avg_sq), ref.exp_avg_sq[3], ref.exp_avg_sq[4], ref.exp_avg_sq[5])
    n = want_p[0].shape[0]
    for name, a, b in zip(NAMES, params, want_p):
        assert a.shape == b.shape, (what, name, a.shape, b.shape)
        if name in ("xyz_", "scaling_") and n_children:
            assert torch.equal(a[:n - n_children], b[:n - n_children]), (what, name, "copied rows")
            tol = dict(rtol=1e-6, atol=1e-6) if name == "xyz_" else dict(rtol=1e-6, atol=2e-7)
            assert torch.allclose(a[n - n_children:], b[n - n_children:], **tol), (what, name, "split children")
        elif not torch.equal(a, b):   # (say where: the first differing row and the size of the difference)
            d = (a - b).abs().reshape(a.shape[0], -1).max(dim=1).values
            row = int(torch.nonzero(d > 0)[0])
            raise AssertionError((what, name, "rows differing", int((d > 0).sum()), "of", a.shape[0], "first", row,
                                  "max abs diff", float(d.max()), a[row].flatten()[:6].tolist(), b[row].flatten()[:6].tolist()))
    for name, a, b in zip(NAMES, m, want_m):
        assert torch.equal(a, b), (what, name, "exp_avg")
    for name, a, b in zip(NAMES, v, want_v):
        assert torch.equal(a, b), (what, name, "exp_avg_sq")
    want_steps = [ref.steps[0], ref.steps[1], ref.steps[3], ref.steps[4], ref.steps[5]]
    assert ref.steps[1] == ref.steps[2] and list(steps) == want_steps, (what, steps, ref.steps)
    for a, b in zip(stats, (ref.accum, ref.denom, ref.max_radii2D)):
        assert a.shape == b.shape and torch.equal(a, b), (what, "statistics")


def same_rows_in_z_order(got, got_z, what):
    """morton_reindex_: the state `got_z` holds the rows of `got` (all five parameters, both moments, the statistics, exist_since_iter_
    -- row for row the same values) in another order, and that order follows a space-filling curve of the positions."""
    def table(st):
        params, m, v, steps, stats, exist = st
        n = params[0].shape[0]
        cols = [t.detach().reshape(n, -1).double() for t in list(params) + list(m) + list(v) + list(stats)] + [exist.reshape(n, 1).double()]
        a = torch.cat(cols, 1).cpu().numpy()
        return a, a[np.lexsort(a.T[::-1])]
    a, sa = table(got)
    z, sz = table(got_z)
    assert a.shape == z.shape and np.array_equal(sa, sz, equal_nan=True), (what, "not the same multiset of rows")
    assert list(got[3]) == list(got_z[3]), (what, "steps")
    step = lambda t: float(np.linalg.norm(np.diff(t[:, :3], axis=0), axis=1).mean())
    assert step(z) < 0.5 * step(a), (what, "rows are not neighbours in space", step(z), step(a))


# (max_grad, min_opacity, extent, max_screen_size, scale spread): the selection cases VERDICT r01 asks for
CASES = {
    "clone_only": dict(max_grad=2e-4, min_opacity=0.0, extent=1e3, mss=0, spread=0.5),     # nothing is "big": clones only
    "split_only": dict(max_grad=2e-4, min_opacity=0.0, extent=1e-3, mss=0, spread=0.5),    # everything is big: splits only
    "mixed": dict(max_grad=2e-4, min_opacity=0.005, extent=None, mss=20, spread=0.7),      # extent = median scale / 0.01
    "prune_all": dict(max_grad=2e-4, min_opacity=1.1, extent=None, mss=0, spread=0.5),     # sigmoid < 1.1 always
    "nothing_selected": dict(max_grad=1e9, min_opacity=0.0, extent=None, mss=0, spread=0.5),
}


def run_case(kind, dev, host_ops, lib_path, name, P=700, seed=0):
    ref_model, ops = _ref_ops(kind)
    c = CASES[name]
    st = make_state(ref_model, P, seed, dev, c["spread"])
    extent = c["extent"]
    if extent is None:
        extent = float(torch.exp(st.params[4]).max(dim=1).values.median()) / 0.01
    seed_rng = 11 + seed
    (torch.cuda.manual_seed if dev.type == "cuda" else torch.manual_seed)(seed_rng)   # the reference draws from the default generator
    ref = ref_model.densify_and_prune(ops, st, 0.01, c["max_grad"], c["min_opacity"], extent, c["mss"])
    n_ref = ref.params[0].shape[0]
    rp._LIB_OVERRIDE = lib_path
    try:
        g = python_host(st, dev)
        gen = torch.Generator(device=dev).manual_seed(seed_rng)
        info = g.densifyAndPrune(c["max_grad"], c["min_opacity"], extent, c["mss"], generator=gen)
        assert info["points"] == n_ref, (name, info, n_ref)
        if name == "clone_only":
            assert info["split"] == 0 and info["cloned"] > 0 and info["pruned"] == 0 and n_ref == P + info["cloned"]
        if name == "split_only":
            assert info["cloned"] == 0 and info["split"] > 0 and n_ref == P + info["split"]
        if name == "mixed":
            assert info["cloned"] > 0 and info["split"] > 0 and info["pruned"] > 0
        if name == "prune_all":
            assert n_ref == 0
        if name == "nothing_selected":
            assert n_ref == P and info["cloned"] == info["split"] == info["pruned"] == 0
        compare(ref, python_state(g), info["children_kept"], name + "/python")
        h = cpp_host(host_ops, st, dev)
        got = list(host_ops.trainer_densify_and_prune(h, c["max_grad"], c["min_opacity"], extent, c["mss"], seed_rng))
        assert got == [info["cloned"], info["split"], info["pruned"], info["points"]], (name, got, info)
        compare(ref, cpp_state(host_ops, h), info["children_kept"], name + "/c++")
        if name in ("mixed", "clone_only", "split_only"):
            # the opt-in Z-order layout (GaussianModel::morton_reindex_): the same rows as the order just checked, permuted
            gz = python_host(st, dev)
            gz.morton_reindex_ = True
            info_z = gz.densifyAndPrune(c["max_grad"], c["min_opacity"], extent, c["mss"], generator=torch.Generator(device=dev).manual_seed(seed_rng))
            assert info_z == info
            same_rows_in_z_order(python_state(g), python_state(gz), name + "/python/morton")
            hz = cpp_host(host_ops, st, dev)
            host_ops.trainer_set_options(hz, {"morton_reindex": 1.0})
            assert list(host_ops.trainer_densify_and_prune(hz, c["max_grad"], c["min_opacity"], extent, c["mss"], seed_rng)) == got
            same_rows_in_z_order(cpp_state(host_ops, h), cpp_state(host_ops, hz), name + "/c++/morton")
            host_ops.trainer_destroy(hz)
        host_ops.trainer_destroy(h)
    finally:
        rp._LIB_OVERRIDE = None


def run_reset_prune_stats(kind, dev, host_ops, lib_path, P=500):
    ref_model, ops = _ref_ops(kind)
    st = make_state(ref_model, P, 3, dev)
    st.params[3][:5, 0] = torch.tensor([20.0, -20.0, 17.5, 0.0, -90.0], device=dev)   # sigmoid saturates: the round trip gives +-inf
    rp._LIB_OVERRIDE = lib_path
    try:
        # --- resetOpacity (:556-565 as shipped: values survive the sigmoid/logit round trip, opacity moments zeroed)
        ref = ref_model.reset_opacity(ops, st)
        g = python_host(st, dev)
        g.resetOpacity()
        compare(ref, python_state(g), 0, "reset/python")
        assert torch.isinf(g.opacity_[0]) and not g.optimizer_.moments(g.opacity_)[0].any()
        finite = torch.isfinite(ref.params[3])
        assert torch.allclose(ref.params[3][finite], st.params[3][finite], rtol=1e-4, atol=1e-5), "the shipped reset keeps the values"
        h = cpp_host(host_ops, st, dev)
        host_ops.trainer_reset_opacity(h)
        compare(ref, cpp_state(host_ops, h), 0, "reset/c++")
        host_ops.trainer_destroy(h)
        # --- prunePoints (:588-642)
        mask = torch.zeros(P, dtype=torch.bool, device=dev)
        mask[::3] = True
        mask[-7:] = True
        ref = ref_model.prune_points(ops, st, mask)
        g = python_host(st, dev)
        g.prunePoints(mask)
        compare(ref, python_state(g), 0, "prune/python")
        h = cpp_host(host_ops, st, dev)
        host_ops.trainer_prune_points(h, mask)
        compare(ref, cpp_state(host_ops, h), 0, "prune/c++")
        host_ops.trainer_destroy(h)
        # --- addDensificationStats (:817-831) against the fused statistics kernel
        gen = torch.Generator().manual_seed(5)
        vs_grad = (1e-3 * torch.randn(P, 3, generator=gen)).to(dev)
        radii = torch.randint(0, 30, (P,), generator=gen).to(dev).int()
        radii[::4] = 0
        vis = radii > 0
        want_acc, want_den = ops.add_densification_stats(st.accum, st.denom, vs_grad, vis)
        g = python_host(st, dev)
        vsp = torch.zeros(P, 3, device=dev, requires_grad=True)
        vsp.grad = vs_grad.clone()
        g.addViewStats(vsp, radii)
        assert torch.allclose(g.xyz_gradient_accum_, want_acc, rtol=1e-6, atol=0) and torch.equal(g.denom_, want_den)
        want_max = st.max_radii2D.clone()
        want_max[vis] = torch.max(want_max[vis], radii[vis].float())     # src/gaussian_mapper.cpp:714-717
        assert torch.equal(g.max_radii2D_, want_max)
    finally:
        rp._LIB_OVERRIDE = None


def run_adam(kind, dev, lib_path, P=300):
    """gsr_adam_step with the group layout / learning rates / eps of trainingSetup against the reference's own
    trainingSetup + torch::optim::Adam::step (C++), three steps in a row."""
    ref_model, ops = _ref_ops(kind)
    st = make_state(ref_model, P, 9, dev)
    gen = torch.Generator().manual_seed(2)
    rp._LIB_OVERRIDE = lib_path
    try:
        g = python_host(st, dev)
        g.spatial_lr_scale_ = 2.5
        opt = GaussianOptimizationParams()
        m = {id(getattr(g, n)): g.optimizer_.state[id(getattr(g, n))] for n in NAMES}
        g.trainingSetup(opt)
        g.optimizer_.state = m
        cur = st
        for it in range(3):
            grads6 = [(1e-3 * torch.randn(*p.shape, generator=gen)).to(dev) for p in cur.params]
            xyz_lr = g.updateLearningRate(it + 1)
            cur = ref_model.adam_step(ops, cur, grads6, spatial_lr_scale=2.5, xyz_lr=xyz_lr)
            for n, gr in zip(NAMES, (grads6[0], torch.cat([grads6[1], grads6[2]], 1), grads6[3], grads6[4], grads6[5])):
                getattr(g, n).grad = gr.clone()
            g.optimizer_.step()
            params, mm, vv, steps, _, _ = python_state(g)
            cat = lambda l: torch.cat([l[1], l[2]], 1)
            for a, b in zip(params, (cur.params[0], cat(cur.params), cur.params[3], cur.params[4], cur.params[5])):
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), it          # <= 1 ulp of the parameter (measured)
            for a, b in zip(mm, (cur.exp_avg[0], cat(cur.exp_avg), cur.exp_avg[3], cur.exp_avg[4], cur.exp_avg[5])):
                assert torch.allclose(a, b, rtol=1e-6, atol=2e-9), it          # 1 ulp of a moment ~1e-2
            for a, b in zip(vv, (cur.exp_avg_sq[0], cat(cur.exp_avg_sq), cur.exp_avg_sq[3], cur.exp_avg_sq[4], cur.exp_avg_sq[5])):
                assert torch.allclose(a, b, rtol=1e-6, atol=2e-11), it
            assert steps == [cur.steps[0]] * 5 == [8 + it] * 5
    finally:
        rp._LIB_OVERRIDE = None


def run_increase_pcd(kind, dev, host_ops, lib_path, P=600, n_new=(157, 40)):
    """GaussianModel::increasePcd (src/gaussian_model.cpp:188-376), both overloads, against both hosts: the new rows (colours ->
    SH DC, kNN scales among the NEW points, identity rotations, opacity 0.1), the carried rows and moments, the zeroed
    statistics, exist_since_iter -- everything bit-equal (the hosts run the same ATen expressions on the same kNN distances; the
    reference's distCUDA2 is the CPU oracle, pinned to simple_knn.cu).  Two insertions in a row (the second one appends in place
    inside the hosts' arena), then a densifyAndPrune that has to carry exist_since_iter through clones and split children."""
    ref_model, ops = _ref_ops(kind)
    st = make_state(ref_model, P, 17, dev, 0.7)
    gen = torch.Generator().manual_seed(23)
    rp._LIB_OVERRIDE = lib_path
    try:
        g = python_host(st, dev)
        h = cpp_host(host_ops, st, dev)
        ref = st
        for k, n in enumerate(n_new):
            pts = (2.0 * torch.randn(n, 3, generator=gen)).to(dev)
            cols = torch.rand(n, 3, generator=gen).to(dev)
            vector_overload = k == 0
            ref = ref_model.increase_pcd(ops, ref, pts, cols, 700 + k, vector_overload)
            assert ref.params[0].shape[0] == P + sum(n_new[:k + 1])
            if vector_overload:
                assert g.increasePcd(pts.cpu().reshape(-1).tolist(), cols.cpu().reshape(-1).tolist(), 700 + k) == n
            else:
                g.increasePcd(pts, cols, 700 + k)
            compare(ref, python_state(g), 0, f"increasePcd {k}/python")
            host_ops.trainer_increase_pcd(h, pts, cols, 700 + k, vector_overload)
            compare(ref, cpp_state(host_ops, h), 0, f"increasePcd {k}/c++")
            assert not ref.accum.any() and not ref.max_radii2D.any()       # densificationPostfix resets the statistics (:709-711)
            assert int(ref.exist_since_iter[-1]) == 700 + k
        # ... and a rebuild on top: the new rows' exist_since_iter_ travels through clone / split / prune
        Pn = ref.params[0].shape[0]
        grads = torch.rand(Pn, 1, generator=gen) * 4e-4
        for obj in (ref,):
            obj.accum, obj.denom = grads.to(dev), torch.ones(Pn, 1, device=dev)
        g.xyz_gradient_accum_, g.denom_ = grads.to(dev).clone(), torch.ones(Pn, 1, device=dev)
        for dst, src in zip(host_ops.trainer_stats(h)[:2], (grads.to(dev), torch.ones(Pn, 1, device=dev))):
            dst.copy_(src)
        extent = float(torch.exp(ref.params[4]).max(dim=1).values.median()) / 0.01
        (torch.cuda.manual_seed if dev.type == "cuda" else torch.manual_seed)(31)
        ref2 = ref_model.densify_and_prune(ops, ref, 0.01, 2e-4, 0.005, extent, 0)
        info = g.densifyAndPrune(2e-4, 0.005, extent, 0, generator=torch.Generator(device=dev).manual_seed(31))
        assert info["cloned"] > 0 and info["split"] > 0
        compare(ref2, python_state(g), info["children_kept"], "increasePcd + densify/python")
        assert list(host_ops.trainer_densify_and_prune(h, 2e-4, 0.005, extent, 0, 31))[3] == ref2.params[0].shape[0]
        compare(ref2, cpp_state(host_ops, h), info["children_kept"], "increasePcd + densify/c++")
        # the arena released: the same values in stable allocations
        host_ops.trainer_release_arena(h)
        compare(ref2, cpp_state(host_ops, h), info["children_kept"], "released arena/c++")
        g.release_arena()
        compare(ref2, python_state(g), info["children_kept"], "released arena/python")
        host_ops.trainer_destroy(h)
    finally:
        rp._LIB_OVERRIDE = None


def _host(variant):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpp_host import load_host
    return load_host(variant)


@pytest.mark.parametrize("name", list(CASES))
def test_densify_and_prune_matches_reference(emu_lib_path, name):
    run_case("cpu", torch.device("cpu"), _host("emu"), emu_lib_path, name)


def test_densify_second_seed_and_size(emu_lib_path):
    run_case("cpu", torch.device("cpu"), _host("emu"), emu_lib_path, "mixed", P=1531, seed=4)


def test_reset_prune_stats_match_reference(emu_lib_path):
    run_reset_prune_stats("cpu", torch.device("cpu"), _host("emu"), emu_lib_path)


def test_increase_pcd_matches_reference(emu_lib_path):
    run_increase_pcd("cpu", torch.device("cpu"), _host("emu"), emu_lib_path)


def test_adam_matches_reference_optimizer(emu_lib_path):
    run_adam("cpu", torch.device("cpu"), emu_lib_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_densify_and_prune_matches_reference_on_gpu(name):
    """the same comparison on the MI355X: the reference's code on HIP tensors (libref_densify_cuda.so: torch::kCUDA is the HIP
    device of a ROCm LibTorch) against both hosts on the HIP kernels, shared Philox seed for at::normal"""
    run_case("cuda", torch.device("cuda:0"), _host("hip"), None, name, P=20000)


@pytest.mark.gpu
def test_reset_prune_stats_adam_match_reference_on_gpu():
    run_reset_prune_stats("cuda", torch.device("cuda:0"), _host("hip"), None, P=20000)
    run_adam("cuda", torch.device("cuda:0"), None, P=20000)


@pytest.mark.gpu
def test_increase_pcd_matches_reference_on_gpu():
    run_increase_pcd("cuda", torch.device("cuda:0"), _host("hip"), None, P=20000, n_new=(5000, 1200))
