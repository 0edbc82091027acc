#!/usr/bin/env python
"""bench.py -- throughput of the hot path on synthetic Gaussian clouds (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config C3]
  (N>1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full training iteration of GaussianMapper::trainForOneIteration on one keyframe
per GPU: render (HIP rasterizer forward) -> masked L1 + 0.2*(1-SSIM) -> backward (HIP
rasterizer backward) -> densification statistics -> [all-reduce of the 6 leaf gradients when
N>1] -> Adam.  Inputs are resident in HBM before the timed region.  N=1 workload = the
configuration BASELINE.json's metric is quoted on: 2M Gaussians at 1920x1080 (config C3).

The JSON line carries, besides the contract fields:
  value            train iterations/s over all GPUs (keyframes optimised per second)
  mpix_per_s       rendered Mpix/s of the rasterizer forward+backward alone (HIP-event time of
                   the rasterizer stages inside the same timed steps)
  roofline         dominant rasterizer kernel: algorithmic bytes / its HIP-event
                   duration vs the 8 TB/s HBM peak; "stages" lists every stage the same way
  cpu_baseline     the CPU oracle (port of the reference kernels) timed on the host cores on
                   the same scene (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(P, V, R, Npix, T, K=16, M=16, tile_passes=2, fused_sh_adam=False):
    """Compulsory HBM bytes per stage (each array read/written once per stage that
    needs it), restated for this implementation's stage split.  fused_sh_adam: the backward preprocess also carries the
    Adam step of the SH tensor (no gradient rows written; both moments read, parameter + moments written for every
    Gaussian, the parameter row read for the culled ones too)."""
    adam = 12 * M * (4 * P + (P - V)) if fused_sh_adam else 0
    return {
        "preprocess_fwd": 52 * P + (12 * K + 67) * V,
        "depth_sort": 4 * 16 * P,                    # 4 passes x (8 B read + 8 B write) over P pairs
        "offset_scan": 12 * P,
        "emit_instances": 20 * P + 8 * R,
        "tile_sort": tile_passes * 16 * R,
        "tile_ranges": 4 * R + 8 * T,
        "blend_fwd": 40 * R + 20 * Npix,
        "grad_memset": R,                            # one flag byte per instance slot
        "blend_bwd": 40 * R + 20 * Npix + 88 * V,
        "preprocess_bwd": 4 * P + 88 * V + (143 + 24 * K) * V + (64 + 12 * M) * (P - V) + adam,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--points", type=int, default=None, help="override the number of Gaussians (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--raster-only", action="store_true", help="time rasterizer fwd+bwd only (no loss/optimizer)")
    ap.add_argument("--host", default="cpp", choices=["cpp", "py"],
                    help="host layer driving the step: the LibTorch C++ one (photo-slam_amd/host, default) or its Python mirror")
    ap.add_argument("--densify-interval", type=int, default=0,
                    help="run densifyAndPrune every N steps inside the timed region (0 = off; reference: 100); with several "
                         "ranks the Python host drives it (it reduces the per-view statistics over the ranks)")
    args = ap.parse_args()
    if args.densify_interval and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        args.host = "py"

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    from photo_slam_amd import capi, scene
    from photo_slam_amd.gaussian_model import GaussianModel, GaussianOptimizationParams
    from photo_slam_amd.gaussian_renderer import GaussianKeyframe, GaussianPipelineParams, GaussianRenderer
    from photo_slam_amd.trainer import TrainStep, GradientReduction, ViewFactoredExchange, FEATURES_GROUP

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # GSR_BENCH_SHARE_GPU=1 + GSR_BENCH_BACKEND=gloo: every rank on device 0 over gloo -- a functional check of the
    # multi-rank path on a 1-GPU box (tools/gpu_dist_smoke.sh); the measured configuration is one rank per GPU over RCCL
    if os.environ.get("GSR_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    # GSR_BENCH_EXCHANGE=allreduce: the plain all-reduce of all five gradient tensors instead of the view-factored exchange
    factored = os.environ.get("GSR_BENCH_EXCHANGE", "factored") != "allreduce"
    # GSR_BENCH_FORCE_DP=1: the data-parallel code path (process group, exchange, per-group Adam) with a single rank -- a
    # functional check of the RCCL calls on a 1-GPU box
    dp = world > 1 or os.environ.get("GSR_BENCH_FORCE_DP") == "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    lib = capi.load()  # raises if libgsr_hip.so is missing

    cfg = scene.CONFIGS[args.config]
    cl = scene.make_config(args.config, seed=0, n_views=max(world, 1), P=args.points)
    cam = cl.cameras[rank % len(cl.cameras)]
    W, H, P = cam.W, cam.H, cl.xyz.shape[0]
    g = GaussianModel.from_cloud(cl, device=dev)
    opt = GaussianOptimizationParams()
    g.trainingSetup(opt)
    kf = GaussianKeyframe.from_camera(cam, dev)
    bg = torch.zeros(3, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    pipe = GaussianPipelineParams()
    # Ground truth = this view of the initial model + low-pass noise: a converged scene under refinement.  The values are
    # not irrelevant to speed: Adam (eps 1e-15) moves every parameter by about its learning rate per step whatever the
    # gradient scale, so fitting a single view drags opacities and scales away from the generated statistics and the blend
    # kernels' work grows step by step (blend_bwd 0.53 -> 0.70 ms over 40 steps with this target, -> 0.80 ms with pure
    # noise).  Throughput is therefore quoted for the default 5 + 20 steps from the generated state.
    noise = torch.nn.functional.avg_pool2d(torch.rand(3, H, W, generator=gen).unsqueeze(0), 5, 1, 2).squeeze(0).to(dev)
    with torch.no_grad():
        gt = (GaussianRenderer.render(kf, H, W, g, pipe, bg)[0] + 0.1 * (noise - 0.5)).clamp_(0.0, 1.0)
    mask = torch.ones(3, H, W, device=dev)
    if args.densify_interval:
        opt.densification_interval_, opt.densify_from_iter_ = args.densify_interval, 0
    ts = TrainStep(g, opt, pipe, bg, world_size=world, cameras_extent=cl.extent,
                   densify=bool(args.densify_interval), factored_exchange=factored)

    ops = None
    if args.host == "cpp" and not args.raster_only:
        sys.path.insert(0, os.path.join(ROOT, "photo-slam_amd", "host"))
        import build_host
        torch.ops.load_library(build_host.build("hip"))
        ops = torch.ops.photoslam_amd
        handle = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                                    g.rotation_.detach(), 3, float(cl.extent), bg)
        import math
        fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
        if dp:
            ops.trainer_set_options(handle, {"fused_sh_adam": 0.0})   # the optimizer follows the gradient exchange
        if dp and factored:
            ops.trainer_set_factored_exchange(handle, True)
        if args.densify_interval:
            ops.trainer_set_options(handle, {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": 0.0,
                                             "densify_from_iter": 0.0, "densification_interval": float(args.densify_interval)})

    # The reference reads the loss on the host every iteration (EMA for logging, gaussian_mapper.cpp:701-705).  So does this
    # loop -- one step late: the value is copied to pinned memory behind the step's kernels and read while the NEXT step is
    # already queued, so the stream never drains for a log value.
    loss_pinned = [torch.zeros(1).pin_memory() for _ in range(2)]
    loss_ready = [torch.cuda.Event() for _ in range(2)]
    loss_state = {"n": 0, "ema": 0.0}

    def read_loss_deferred(loss):
        k = loss_state["n"] & 1
        if loss_state["n"] >= 1:
            loss_ready[k ^ 1].synchronize()                       # step n-1: finished long ago, or being finished now
            loss_state["ema"] = 0.4 * float(loss_pinned[k ^ 1][0]) + 0.6 * loss_state["ema"]
        loss_pinned[k].copy_(loss.detach().reshape(1), non_blocking=True)
        loss_ready[k].record()
        loss_state["n"] += 1

    def one_step():
        if ops is not None:
            loss = ops.trainer_render_and_backward(handle, kf.world_view_transform_, kf.full_proj_transform_,
                                                   kf.camera_center_, fovx, fovy, H, W, gt, mask)
            if dp and factored:
                # view-factored exchange (trainer.ViewFactoredExchange): the colour gradients are gathered, the other four
                # tensors reduced; the SH gradient is rebuilt from the views and applied while the reductions are on the links
                grads = ops.trainer_grads(handle)
                ex = ViewFactoredExchange(ops.trainer_sh_send_buffer(handle), kf.camera_center_,
                                          [(i, t) for i, t in enumerate(grads) if i != FEATURES_GROUP], world)
                ops.trainer_finish_begin(handle)
                centres, views = ex.gathered()
                ops.trainer_features_step_from_views(handle, centres, views)   # rebuild + Adam in one pass; reads xyz: before ITS Adam
                for i in ex.order():
                    ex.wait(i)                # stream-side wait: the host keeps queueing
                    ops.trainer_adam_group(handle, i)
                ops.trainer_finish_end(handle)
            elif dp:
                # reductions in flight from here (largest first); each tensor's Adam follows ITS reduction, so the
                # SH update overlaps the small reductions still on the links
                red = GradientReduction(ops.trainer_grads(handle), world)
                ops.trainer_finish_begin(handle)
                for i in red.order():
                    red.wait(i)               # stream-side wait: the host keeps queueing
                    ops.trainer_adam_group(handle, i)
                ops.trainer_finish_end(handle)
            else:
                ops.trainer_finish(handle)    # statistics + Adam
            read_loss_deferred(loss)
        elif args.raster_only:
            img, vsp, vis, radii = GaussianRenderer.render(kf, H, W, g, pipe, bg)
            img.backward(gt)
            g.optimizer_.zero_grad(set_to_none=True)
        else:
            ts.trainForOneIteration(kf, gt, mask, sync_loss=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    # Timed region: HIP events only around the backward blend, the dominant kernel (gsr_profile_enable(2)): every event
    # record is a ~5 us bubble in the stream, and eleven of them per step cost 2 % of the step they are meant to measure.
    capi.profile_enable(lib, 2)
    dom_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        v = capi.profile_read(lib)["blend_bwd"]      # waits only for events already recorded this step
        if v >= 0:
            dom_ms.append(v)
    barrier()
    elapsed = time.perf_counter() - t0
    # Stage table: the same step with events between all stages, outside the timed region -- and with the separate Adam pass
    # on the SH tensor, so that the rasterizer stages (and the Mpix/s derived from them) contain no optimizer work.
    if ops is not None:
        ops.trainer_set_options(handle, {"fused_sh_adam": 0.0})
    ts.fused_sh_adam_ = False
    capi.profile_enable(lib, 1)
    stage_ms = {}
    for _ in range(min(args.steps, 10)):
        one_step()
        for k, v in capi.profile_read(lib).items():
            stage_ms.setdefault(k, []).append(v)
    torch.cuda.synchronize()
    capi.profile_enable(lib, 0)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # scene statistics of this rank's view (V, R) for the byte model
    with torch.no_grad():
        img, _, vis, radii = GaussianRenderer.render(kf, H, W, g, pipe, bg)
        V = int(vis.sum().item())
    from photo_slam_amd import rasterize_points as rp  # noqa
    # R: instances of the last forward = sum of tiles; recompute through the public wrapper
    R = rp.RasterizeGaussiansCUDA(bg, g.getXYZ().detach(), torch.empty(0, device=dev), g.getOpacityActivation().detach(),
                                  g.getScalingActivation().detach(), g.getRotationActivation().detach(), 1.0,
                                  torch.empty(0, device=dev), kf.world_view_transform_, kf.full_proj_transform_,
                                  kf.tanfovx_, kf.tanfovy_, H, W, g.getFeatures().detach(), 3, kf.camera_center_, False)[0]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    tile_bits = int(np.ceil(np.log2(max(T, 2))))
    fused_sh_adam = not dp and not args.raster_only   # both hosts fuse the SH Adam step into backward at one rank (timed region)
    ab = algorithmic_bytes(P, V, R, W * H, T, tile_passes=(tile_bits + 7) // 8)   # the stage table ran unfused (above)
    stages = {}
    for k, ms in stage_ms.items():
        ms = [m for m in ms if m >= 0]
        if not ms:
            continue
        avg = float(np.mean(ms))
        stages[k] = dict(ms=round(avg, 4), bytes=int(ab[k]), GBps=round(ab[k] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None)
    raster_ms = sum(s["ms"] for s in stages.values())
    dom = max(stages, key=lambda k: stages[k]["ms"]) if stages else None
    if dom == "blend_bwd" and dom_ms:
        # the dominant kernel's duration inside the timed region replaces the stage-table value
        avg = float(np.mean(dom_ms))
        stages[dom] = dict(ms=round(avg, 4), bytes=int(ab[dom]), GBps=round(ab[dom] / (avg * 1e-3) / 1e9, 1),
                           timed_region=True)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "train iters/s (render + L1/SSIM loss + backward + Adam), 2M Gaussians @1080p"
            if args.config == "C3" else f"train iters/s, config {args.config}",
            "value": round(world * args.steps / elapsed, 3),
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['note']}", "gaussians": P, "width": W, "height": H,
                       "visible": V, "instances": R, "keyframes_per_step": world, "sh_degree": 3,
                       "parallelism": "single GPU" if not dp else
                                      (f"dp{world} (one keyframe per GPU; one all-gather of 3 + one all-reduce of 11 floats/Gaussian, "
                                       "SH gradient rebuilt per rank; densification statistics accumulate per rank)") if factored else
                                      f"dp{world} (one keyframe per GPU, all-reduce of 59 floats/Gaussian)",
                       "raster_only": bool(args.raster_only), "densify_interval": args.densify_interval,
                       "sh_adam_fused_into_backward": fused_sh_adam,
                       "gaussians_after": int(g.xyz_.shape[0]) if ops is None else P,
                       "host": "libtorch-c++ (photo-slam_amd/host)" if ops is not None else "python mirror"},
            "mpix_per_s": round(world * W * H / (raster_ms * 1e-3) / 1e6, 1) if raster_ms > 0 else None,
            "raster_fwd_bwd_ms": round(raster_ms, 4),
        }
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_j_pmc_traffic_C3_raster_only.json")
        # HBM bytes per launch from rocprofv3 TCC counters (separate --pmc passes, tools/gpu_pmc.sh), corrected as
        # MI355X_MICROARCH.md prescribes for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Measured for C3 only.
        kernel_of = {"blend_bwd": "gsr::blend_bwd_kernel", "blend_fwd": "gsr::blend_fwd_kernel",
                     "preprocess_fwd": "gsr::preprocess_fwd_kernel", "preprocess_bwd": "gsr::preprocess_bwd_kernel"}
        if dom and args.config == "C3" and args.points is None and dom in kernel_of and os.path.exists(pmc_file):
            pm = json.load(open(pmc_file)).get(kernel_of[dom])
            if pm:
                traffic = int((2 * pm.get("FETCH_SIZE", 0) + pm.get("WRITE_SIZE", 0)) * 1024)
        # the blend kernels are VALU-bound: their VALU issue utilisation from the SQ counters (tools/gpu_sq.sh, C3 only) rides along
        valu = None
        sq_file = os.path.join(ROOT, "profiles", "r01_j_sq_counters_C3_raster_only.json")
        if dom and args.config == "C3" and args.points is None and dom in kernel_of and os.path.exists(sq_file):
            sq = json.load(open(sq_file)).get(kernel_of[dom])
            if sq and "valu_issue_utilisation" in sq:
                valu = round(sq["valu_issue_utilisation"], 3)
        if dom:
            a = stages[dom]["GBps"]
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic, "valu_issue_utilisation": valu,
                               "raster_fwd_bwd_frac": round(sum(ab.values()) / (raster_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "stages": stages}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            t1 = time.perf_counter()
            ores, ocolor, oradii = oracle.forward(np.zeros(3, np.float32), cl.xyz, cl.get_opacity(), cam.viewmatrix,
                                                  cam.projmatrix, cam.campos, cam.tanfovx, cam.tanfovy, H, W,
                                                  shs=cl.get_features(), sh_degree=3, scales=cl.get_scaling(),
                                                  rotations=cl.get_rotation())
            oracle.backward(ores, np.ones((3, H, W), np.float32))
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(W * H / cpu_s / 1e6, 3), "unit": "Mpix/s (rasterizer fwd+bwd)",
                                   "cores": oracle.get_threads(), "kind": "port",
                                   "sample": f"1 forward+backward of the same {args.config} view (initial parameters), "
                                             f"{cpu_s:.1f} s, OpenMP over Gaussians/tiles",
                                   "iters_per_s_raster_only": round(1.0 / cpu_s, 4)}
    if dp:
        dist.destroy_process_group()
    # RCCL writes its version banner to C stdout, which is block-buffered on a pipe and would otherwise land AFTER the
    # result at exit: drain it first, so that the JSON line is the last line of the output
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
