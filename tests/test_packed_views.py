"""The PACKED exchange of the colour gradients (include/gsr.h: gsr_pack_color_view, gsr_sh_grad_from_packed_views,
gsr_sh_adam_from_packed_views): a rank sends the rows its view SEES -- mask + packed rows + camera centre -- instead of the
[P + 1, 3] buffer.  The rebuilt SH gradient and the fused (lazy) Adam step must be BIT-IDENTICAL to the dense form: the same
rows, the same order of the views.  Cases: every view sees about half the cloud; one view sees nothing; one sees everything; a
capacity that is exactly K; P not a multiple of 64; the -0.0f visibility marker; and gsr_last_visible_count()."""
import numpy as np
import pytest
import torch

from photo_slam_amd import rasterize_points as rp
from photo_slam_amd import scene


def _views(P, n_views, dev, seed, fractions):
    g = torch.Generator().manual_seed(seed)
    views = torch.zeros(n_views, P, 3)
    for v, frac in enumerate(fractions):
        seen = torch.rand(P, generator=g) < frac
        rows = 1e-3 * torch.randn(P, 3, generator=g)
        unlit = torch.rand(P, generator=g) < 0.1          # visible, every channel zero: the marker -0.0f in channel 0
        rows[unlit] = torch.tensor([-0.0, 0.0, 0.0])
        views[v][seen] = rows[seen]
    centres = torch.randn(n_views, 3, generator=g) * 2.0
    return views.to(dev), centres.to(dev)


def run_packed_checks(dev, lib_path, P=1000, n_views=3, fractions=(0.5, 0.45, 0.55), seed=0):
    rp._LIB_OVERRIDE = lib_path
    try:
        g = torch.Generator().manual_seed(seed + 100)
        means = (torch.randn(P, 3, generator=g) * 2.0).to(dev)
        views, centres = _views(P, n_views, dev, seed, fractions)
        counts = [int(((views[v].view(torch.int32) != 0).any(1)).sum()) for v in range(n_views)]
        capacity = (max(counts) + 3) // 4 * 4                 # what the ranks agree on: max K, a multiple of 4
        words = rp.packedViewWords(P, capacity)
        msgs = torch.empty(n_views, words, dtype=torch.int32, device=dev)
        for v in range(n_views):
            rp.packColorView(views[v], centres[v], capacity, msgs[v])
            hdr = msgs[v][:8].cpu()
            assert int(hdr[0]) == counts[v] and int(hdr[1]) == P and int(hdr[2]) == capacity and int(hdr[3]) == 0
            assert torch.equal(hdr[4:7].view(torch.float32), centres[v].cpu())
        # the rebuilt gradient
        dense = rp.shGradFromViews(means, centres, views, 3, 16, 1.0 / n_views)
        packed = rp.shGradFromPackedViews(means, msgs, words, n_views, 3, 16, 1.0 / n_views)
        assert torch.equal(dense, packed)
        assert bool(dense.abs().sum() > 0)
        # the fused Adam step, eager and lazy
        for lazy in (False, True):
            sh0 = torch.randn(P, 16, 3, generator=g).to(dev)
            m0 = (0.01 * torch.randn(P, 16, 3, generator=g)).to(dev)
            v0 = (1e-4 * torch.rand(P, 16, 3, generator=g)).to(dev)
            out = []
            for packed_form in (False, True):
                sh, m, v = sh0.clone(), m0.clone(), v0.clone()
                adam = dict(exp_avg=m, exp_avg_sq=v, lr=0.0025, lr_tail=0.0025 / 20, beta1=0.9, beta2=0.999, eps=1e-15, step=5)
                if lazy:
                    row_step = torch.randint(2, 5, (P,), generator=torch.Generator().manual_seed(7), dtype=torch.int32).to(dev)
                    adam.update(row_step=row_step, window=4, lr_past=[0.0025] * 3, lr_tail_past=[0.0025 / 20] * 3)
                if packed_form:
                    rp.shAdamFromPackedViews(means, msgs, words, n_views, 3, 1.0 / n_views, sh, adam)
                else:
                    rp.shAdamFromViews(means, centres, views, 3, 1.0 / n_views, sh, adam)
                out.append((sh, m, v, adam.get("row_step")))
            for a, b in zip(out[0], out[1]):
                if a is not None:
                    assert torch.equal(a, b)
            assert not torch.equal(out[0][0], sh0)
        # the gathered headers read back: these messages describe P rows at this capacity, nothing dropped
        rp.checkPackedViews(msgs, words, n_views, P, capacity)
        # a capacity below K: the overflow flag is raised (a caller's bug made visible, never a silent truncation)
        small = max(capacity - 8, 0)
        worst = int(np.argmax(counts))
        over = rp.packColorView(views[worst], centres[0], small)
        assert int(over[3].cpu()) == 1 and int(over[0].cpu()) == max(counts)
        if max(counts) > small:
            # ... the check refuses such a message, and the decoder reads no row beyond the ones it holds (rows past `small`
            # decode as zero instead of as the words behind the message)
            with pytest.raises(RuntimeError, match="dropped rows"):
                rp.checkPackedViews(over, over.numel(), 1, P, small)
            got = rp.shGradFromPackedViews(means, over, over.numel(), 1, 3, 16, 1.0)
            seen = (views[worst].view(torch.int32) != 0).any(1)
            rank = torch.cumsum(seen.to(torch.int64), 0) - 1
            held = seen & (rank < small)
            cut = views[worst].clone()
            cut[~held] = 0.0
            cut_msg = rp.packColorView(cut, centres[0], small)
            assert torch.equal(got, rp.shGradFromPackedViews(means, cut_msg, cut_msg.numel(), 1, 3, 16, 1.0))
        # a message written for another P is not decoded at all, and the check says so
        alien = msgs.clone()
        alien[0][1] = P + 64
        with pytest.raises(RuntimeError, match="does not describe"):
            rp.checkPackedViews(alien, words, n_views, P, capacity)
        if n_views > 1:
            rest = rp.shGradFromPackedViews(means, msgs[1:].contiguous(), words, n_views - 1, 3, 16, 1.0 / n_views)
            assert torch.equal(rp.shGradFromPackedViews(means, alien, words, n_views, 3, 16, 1.0 / n_views), rest)
    finally:
        rp._LIB_OVERRIDE = None


def run_visible_count_check(dev, lib_path):
    import parity
    cl = scene.make_cloud(3000, 96, 64, 80.0, 80.0, seed=4, scale_k=0.3)
    r = parity.run_backend(lib_path, dev, cl, cl.cameras[0], np.zeros(3, np.float32), do_backward=False)
    rp._LIB_OVERRIDE = lib_path
    try:
        assert rp.lastVisibleCount() == int((r.radii > 0).sum()) > 0
    finally:
        rp._LIB_OVERRIDE = None


def run_backward_packs_check(dev, lib_path, cl, sh_degree=3):
    """gsr_backward_args.packed_view: the backward pass writes the view's message itself (mask + prefix planned from the radii by
    gsr_pack_view_plan) -- the same words as gsr_pack_color_view on the dense view it leaves next to it, and nothing else changes."""
    import parity
    bg = np.zeros(3, np.float32)
    for k, cam in enumerate(cl.cameras):
        dpix = np.random.default_rng(k).standard_normal((3, cam.H, cam.W)).astype(np.float32)
        a = parity.run_backend(lib_path, dev, cl, cam, bg, sh_degree=sh_degree, dL_dpix=dpix, factored=True)
        b = parity.run_backend(lib_path, dev, cl, cam, bg, sh_degree=sh_degree, dL_dpix=dpix, factored=True, packed=True)
        for n in a.grads:
            if dev.type == "cpu":
                assert np.array_equal(a.grads[n], b.grads[n], equal_nan=True), n
            else:   # (two passes on the GPU differ by the order of the four LDS adds that merge a tile's quads)
                d = np.abs(a.grads[n].astype(np.float64) - b.grads[n]).sum() / max(np.abs(a.grads[n]).sum(), 1e-30)
                assert d < 1e-5, (n, d)
        view = torch.from_numpy(b.grads["dL_dcolor_view"]).to(dev)
        P, cap = view.shape[0], b.packed_capacity
        rp._LIB_OVERRIDE = lib_path
        try:
            blank = torch.full((rp.packedViewWords(P, cap),), -1, dtype=torch.int32, device=dev)   # (as parity.run_backend: padding words stay)
            want = rp.packColorView(view, torch.from_numpy(np.ascontiguousarray(cam.campos)).to(dev), cap, blank).cpu()
        finally:
            rp._LIB_OVERRIDE = None
        K = int(want[0])
        assert K == int((b.radii > 0).sum()) > 0 and int(want[3]) == 0
        used = rp.packedViewWords(P, (K + 3) // 4 * 4) - 4 - 3 * ((K + 3) // 4 * 4 - K)     # header + prefix + masks + K rows
        assert torch.equal(b.packed_msg[:used], want[:used]), f"camera {k}: the message of the backward pass differs"


def test_backward_writes_the_packed_message_on_the_emulator(emu_lib_path):
    cl = scene.make_cloud(3000, 96, 64, 80.0, 80.0, seed=4, scale_k=0.3, n_views=2)
    run_backward_packs_check(torch.device("cpu"), emu_lib_path, cl)
    cl = scene.make_cloud(130, 48, 32, 40.0, 40.0, seed=5, scale_k=0.4)      # P not a multiple of 64, a last group of 2 rows
    run_backward_packs_check(torch.device("cpu"), emu_lib_path, cl, sh_degree=1)


@pytest.mark.gpu
def test_backward_writes_the_packed_message_on_gpu():
    run_backward_packs_check(torch.device("cuda:0"), None, scene.make_config("C1", seed=0, n_views=2))
    run_backward_packs_check(torch.device("cuda:0"), None, scene.make_config("C2", seed=1))


def test_packed_views_equal_the_dense_exchange_on_the_emulator(emu_lib_path):
    dev = torch.device("cpu")
    run_packed_checks(dev, emu_lib_path)
    run_packed_checks(dev, emu_lib_path, P=777, n_views=4, fractions=(0.0, 1.0, 0.3, 0.5), seed=1)    # nothing / everything seen
    run_packed_checks(dev, emu_lib_path, P=64, n_views=1, fractions=(0.5,), seed=2)
    run_packed_checks(dev, emu_lib_path, P=5000, n_views=8, fractions=(0.47,) * 8, seed=3)
    run_visible_count_check(dev, emu_lib_path)


@pytest.mark.gpu
def test_packed_views_equal_the_dense_exchange_on_gpu():
    dev = torch.device("cuda:0")
    run_packed_checks(dev, None, P=200_003, n_views=4, fractions=(0.5, 0.4, 0.0, 1.0), seed=5)
    run_packed_checks(dev, None, P=1_000_000, n_views=8, fractions=(0.47,) * 8, seed=6)
    run_visible_count_check(dev, None)
