#!/usr/bin/env python
"""bench.py -- throughput of the hot path on synthetic Gaussian clouds (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config C3]
  (N > 1 without WORLD_SIZE in the environment: re-executes itself under torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full training iteration of GaussianMapper::trainForOneIteration (src/gaussian_mapper.cpp:614-774) on one
keyframe per GPU: render (HIP rasterizer forward) -> masked L1 + 0.2*(1-SSIM) -> backward (HIP rasterizer backward) ->
densification statistics -> [gradient exchange when N>1] -> Adam.  Inputs are resident in HBM before the timed region.
N=1 workload = the configuration BASELINE.json's metric is quoted on: 2M Gaussians at 1920x1080 (config C3).

Protocol:
  * `value` / `ms_per_step`: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize (driver contract).
  * The workload is STATIONARY: the timed legs run with every learning rate multiplied by 0 (`config.learning_rates`) -- the
    same kernels, the same bytes, the Adam moments update, the parameters do not move.  With the training learning rates
    the synthetic scene inflates by ~1 %/step (fitting one noisy view with Adam's eps 1e-15 drags opacities and scales
    away from the generated statistics), so a mean over the first 20 steps flattered the number; `training_lr_run` reports
    that run too, K steps from the same start.
  * `protocol`: per-step times from HIP events on the stream, 20 warm-up + median of >= 100 steps, next to the wall clock.
  * `roofline`: the dominant rasterizer kernel's duration from HIP events INSIDE the timed region; the stage table comes
    from the same (fused) program, recorded in further steps with events between all stages.
  * `cpu_baseline`: the reference's train step on the host cores (oracle/cpu_trainer.py), rank 0 at N=1 only.
"""
import argparse
import json
import os
import socket
import sys
import time

# the CPU baseline leg alternates between two OpenMP consumers (the oracle and ATen): spinning worker threads of the one
# starve the other on a many-core host -- sleep instead (must be set before libgomp loads)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# committed counter profiles of the same build (tools/gpu_pmc.sh, tools/gpu_sq.sh): newest set first (names sort by round + tag)
import glob
PMC_FILES = sorted((os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_C3_raster_only.json"))),
                   reverse=True)


SQ_FILES = sorted((os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters_full.json"))), reverse=True)
ISSUE_COST_FILES = sorted((os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_blend_issue_cost.json"))),
                          reverse=True)


def sq_probe(config, seed=0):
    """VALU issue of the two blend kernels MEASURED on this box: one `rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE` pass (counters
    only, next to --kernel-trace, as the pool's rules ask) over a short run of this program at `config` -> per kernel the mean
    counter values per launch, or (None, reason) when the profiler is not on the box / the pass fails."""
    import collections
    import csv
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 is not on PATH"
    d = tempfile.mkdtemp(prefix="gsr_sq_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__), "--config", config, "--seed", str(seed), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--median-steps", "0", "--densify-leg-steps", "0", "--no-knn-leg", "--dropin-steps", "0", "--no-config-legs", "--no-sq-probe"]
    try:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, check=True)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None, "the counter pass left no counter_collection.csv"
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if "gsr::blend_" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
        for k in out:
            out[k]["launches"] = len(next(iter(acc[k].values())))
        return (out, None) if out else (None, "no blend kernel in the counter file")
    except Exception as e:   # (a box without counter access, a timeout: the committed profile is used and the record says so)
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def config_leg(config, seed, steps=40, warmup=10, extra=()):
    """One more single-GPU BASELINE config in THIS record: the same program at `config` in a child process (quick form: no CPU /
    kNN / densify / drop-in legs), reduced to its step time, the rasterizer alone, its own roofline entry and the blend kernels'
    VALU issue utilisation measured there."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--seed", str(seed), "--steps", str(steps), "--warmup", str(warmup),
           "--median-steps", str(steps), "--no-cpu-baseline", "--no-knn-leg", "--densify-leg-steps", "0", "--dropin-steps", "0", "--no-config-legs"]
    cmd += list(extra)
    t0 = time.time()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith('{"metric"')), None)
        if r.returncode != 0 or line is None:
            return {"config": config, "error": (r.stderr or "no JSON line")[-300:]}
        d = json.loads(line)
    except Exception as e:
        return {"config": config, "error": f"{type(e).__name__}: {e}"[:300]}
    rf = d.get("roofline", {})
    return {"workload": d["config"]["workload"], "gaussians": d["config"]["gaussians"], "width": d["config"]["width"], "height": d["config"]["height"],
            "visible": d["config"]["visible"], "instances": d["config"]["instances"], "binning": d["config"].get("binning"),
            "iters_per_s": d["value"], "ms_per_step": d["ms_per_step"], "median_ms_per_step": d.get("protocol", {}).get("median_ms_per_step"),
            "steps": d["steps"], "warmup": d["warmup"], "rasterizer_only": d.get("rasterizer_only"),
            "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "valu_issue_utilisation", "traffic",
                                                "raster_fwd_bwd_frac", "bounds")},
            "stages_ms": {k: v["ms"] for k, v in rf.get("stages", {}).items()}, "leg_wall_s": round(time.time() - t0, 1)}


def stage_bounds(stages, top=6, sq_live=None, sq_live_note=None):
    """What bounds each of the `top` largest stages, so that the claim rides in this record and not only in DESIGN.md:
      * the two blend kernels: "valu-issue" -- SIMD cycles available per wave-VALU instruction issued (SQ counters of the newest
        committed profile of this program: GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs / SQ_INSTS_VALU) against the mean issue cost of
        the visit loop's instruction mix (tools/isa_cost.py on the compiler's ISA, weighted with tools/valu_rate.hip's measured
        cycles per instruction class); utilisation = cost / available.  Read from committed files, not measured in this run;
      * the streaming stages: "hbm" with achieved = algorithmic bytes / stage time of THIS run against the 8 TB/s peak;
      * the sorts and scans: "launch-latency" (6-12 launches of 5-20 us each) with the same achieved figure for reference."""
    sq = json.load(open(os.path.join(ROOT, SQ_FILES[0]))) if SQ_FILES else {}
    sq_src = SQ_FILES[0] + " (committed profile of this program: NOT measured in this run" + (f" -- {sq_live_note})" if sq_live_note else ")") if SQ_FILES else None
    if sq_live:   # (sq_probe: measured on this box, in this run)
        sq, sq_src = sq_live, "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE over 3 steps of this program on this box"
    cost = json.load(open(os.path.join(ROOT, ISSUE_COST_FILES[0]))) if ISSUE_COST_FILES else {}
    out = {}
    for k in sorted(stages, key=lambda k: -stages[k]["ms"])[:top]:
        st = stages[k]
        frac = round(st["GBps"] / HBM_PEAK_GBS, 4) if st.get("GBps") else None
        if k in ("blend_fwd", "blend_bwd"):
            c = sq.get(f"gsr::{k}_kernel")
            e = {"bound": "valu-issue", "ms": st["ms"], "hbm_frac_of_algorithmic_bytes": frac}
            if c and c.get("SQ_INSTS_VALU"):
                avail = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / c["SQ_INSTS_VALU"]
                e["simd_cycles_per_valu_instruction"] = round(avail, 3)
                e["sq_counters_source"] = sq_src
                if k in cost:
                    e["issue_cost_of_the_instruction_mix"] = cost[k]["mean_cycles_per_valu"]
                    e["valu_issue_utilisation"] = round(cost[k]["mean_cycles_per_valu"] / avail, 3)
                    e["issue_cost_source"] = f"{ISSUE_COST_FILES[0]} ({cost.get('cost_source', '')})"
            out[k] = e
        elif k in ("depth_sort", "tile_sort", "offset_scan"):
            out[k] = {"bound": "launch-latency", "ms": st["ms"], "hbm_frac_of_algorithmic_bytes": frac}
        else:
            out[k] = {"bound": "hbm", "ms": st["ms"], "achieved_GBps": st.get("GBps"), "frac_of_8TBps": frac}
    return out


def algorithmic_bytes(P, V, R, Npix, T, K=16, M=16, tile_passes=2, depth_passes=4, sorted_gaussians=None, fused_sh_adam=False,
                      touched_slots=None, lazy_window=0, fused_geom_adam=False, tile_first=False):
    """Compulsory HBM bytes per stage (each array read/written once per stage that needs it), restated for
    this implementation's stage split.  fused_sh_adam: the backward preprocess also carries the Adam step of the SH tensor
    (no gradient rows written; both moments read, parameter + moments written for every Gaussian, the parameter row read for
    the culled ones too).  touched_slots: instance slots the backward blend wrote (48 B each, read back once by
    preprocess_bwd together with R flag bytes); None leaves them out.  lazy_window: the culled rows take their zero-gradient
    steps `lazy_window` at a time (gsr_sh_adam_lazy): per step, moments read + parameter and moments written for the VISIBLE
    rows, and a full read-modify-write of 1/lazy_window of the culled ones.  fused_geom_adam: the stage also carries the Adam
    steps of xyz / opacity / scaling / rotation (11 floats per Gaussian: parameter and two moments read and written, 264 B)
    and no longer writes their gradients, the viewspace gradient and dL_dcov3D (80 B per Gaussian)."""
    if fused_sh_adam and lazy_window >= 2:
        adam = 12 * M * (5 * V + 6 * (P - V) // lazy_window - P)   # (- P: the stage's gradient rows are not written)
    else:
        adam = 12 * M * (4 * P + (P - V)) if fused_sh_adam else 0
    if fused_geom_adam:
        adam += (264 - 80) * P
    S = P if sorted_gaussians is None else sorted_gaussians
    slots = 0 if touched_slots is None else 48 * touched_slots + R
    return {
        "preprocess_fwd": 52 * P + (12 * K + 67) * V,
        # (tile-first binning: this stage is the compaction -- tile counts read, id / offset / rectangle of the visible ones written)
        "depth_sort": 4 * P + 24 * V if tile_first else depth_passes * 16 * S,   # passes x (8 B read + 8 B write) over the sorted pairs
        "offset_scan": 0 if tile_first else 12 * S,
        "emit_instances": 20 * S + 8 * R,
        "tile_sort": tile_passes * 16 * R,
        "tile_ranges": 4 * R + 8 * T,
        "tile_depth_sort": 12 * R if tile_first else 0,   # tile-first binning: list entry read, depth key gathered, entry written
        "blend_fwd": 40 * R + 20 * Npix,
        "grad_memset": R,                            # one flag byte per instance slot
        "blend_bwd": 40 * R + 20 * Npix + 88 * V,
        "preprocess_bwd": 4 * P + 88 * V + (143 + 24 * K) * V + (64 + 12 * M) * (P - V) + adam + slots,
    }


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become `python -m torch.distributed.run ... bench.py ...`."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def cpu_train_step_baseline(scene, args, W, H, quick=False, want_inputs=False):
    """The reference's CPU train step (oracle/cpu_trainer.py) on the box's host cores: C1 in full (>= 100 iterations after
    warm-up) and >= 3 measured iterations of the benchmarked configuration.  ~20-30 s of CPU work on the GPU box."""
    from oracle import cpu_trainer, oracle
    out = {}
    keep = {}   # per config: the cloud, the ground truth and every loss of the CPU run (for the GPU loss check; not printed)

    def run(name, iterations, warmup, budget_s):
        cl = scene.make_config(name, seed=0)
        cam = cl.cameras[0]
        res, color, _ = oracle.forward(np.zeros(3, np.float32), cl.xyz, cl.get_opacity(), cam.viewmatrix, cam.projmatrix, cam.campos,
                                       cam.tanfovx, cam.tanfovy, cam.H, cam.W, shs=cl.get_features(), sh_degree=3,
                                       scales=cl.get_scaling(), rotations=cl.get_rotation())
        res.free()
        rng = np.random.default_rng(1234)
        gt = np.clip(color + 0.1 * (rng.random(color.shape, dtype=np.float32) - 0.5), 0.0, 1.0)   # as the GPU run: this view + noise
        t0 = time.perf_counter()
        r = cpu_trainer.train(cl, cam, gt, iterations, warmup=warmup, time_budget_s=budget_s)
        med = float(np.median(r["seconds"]))
        keep[name] = dict(cloud=cl, gt=gt, losses=list(r["losses"]))
        return dict(config=name, gaussians=int(cl.xyz.shape[0]), width=cam.W, height=cam.H, iterations=len(r["seconds"]),
                    iterations_requested=iterations, time_budget_s=budget_s, warmup=warmup,
                    s_per_iteration_median=round(med, 4), iters_per_s=round(1.0 / med, 4),
                    mpix_per_s_fwd_bwd=None, wall_s=round(time.perf_counter() - t0, 1), loss_first=round(r["losses"][0], 5),
                    loss_last=round(r["losses"][-1], 5), loss_ops=r["loss_ops"], torch_threads=r["threads"],
                    phase_s_median=dict(zip(("render_forward", "loss_forward", "backward", "stats_and_adam"), r["phase_seconds_median"])),
                    oracle_threads=r["oracle_threads"])
    out["C1"] = run("C1", 20 if quick else 100, 2 if quick else 5, 10.0 if quick else 80.0)
    main_cfg = args.config if args.config != "C1" else None
    if main_cfg:
        out[main_cfg] = run(main_cfg, 3, 1, 60.0)
    if want_inputs:
        return out, keep
    return out


def dropin_unfused_leg(torch, dev, cl, kf, fovx, fovy, H, W, gt, steps, stationary=True, fused_loss=False):
    """What a maintainer gets from the API-only swap: the REFERENCE's own host code -- src/gaussian_trainer.cpp:45-133 calling
    src/gaussian_renderer.cpp / src/gaussian_rasterizer.cpp, ATen activations, cat(dc.clone(), rest.clone()), include/loss_utils.h
    through autograd (MIOpen convolutions), torch::optim::Adam, addDensificationStats -- compiled UNCHANGED
    (oracle/build_ref.py: build_host_tree -> oracle/_ref/libref_host_hip.so, prebuilt: the reference tree does not exist on the
    GPU box) and linked against lib/libcuda_rasterizer.so + lib/libsimple_knn.so only.  Same cloud, same keyframe, same ground
    truth as `value`; a baseline beside it, never `value`.  Every iteration ends in torch::cuda::synchronize() + loss.item() as
    the reference's loop does (:86,92), so wall-clock differences are step times: two calls of GaussianTrainer::trainingOnce
    with n and n + steps iterations, (t2 - t1) / steps."""
    # fused_loss: the same unchanged sources compiled with ONE header swapped -- include/loss_utils.h -> this repository's
    # host/include/loss_utils.h (same names and signatures; l1_loss / ssim on the fused HIP kernels): INTEGRATION.md section 5
    name = "libref_host_hip_fused_loss.so" if fused_loss else "libref_host_hip.so"
    path = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(path):
        return {"skipped": f"oracle/_ref/{name} was never built (python __graft_entry__.py where /root/reference exists)"}
    torch.ops.load_library(path)
    rops = torch.ops.photoslam_reference_host_fl if fused_loss else torch.ops.photoslam_reference_host
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    h = rops.create([t(cl.xyz), t(cl.features_dc), t(cl.features_rest), t(cl.opacity), t(cl.scaling), t(cl.rotation)], 3, 3,
                    float(cl.extent), float(cl.extent))
    rops.add_keyframe(h, 0, kf.world_view_transform_, kf.full_proj_transform_, kf.camera_center_, fovx, fovy, H, W, gt)
    never = 1.0e9
    opts = {"densify_from_iter": never, "opacity_reset_interval": never}    # statistics every iteration, no densification
    if stationary:
        opts.update({k: 0.0 for k in ("position_lr_init", "position_lr_final", "feature_lr", "opacity_lr", "scaling_lr", "rotation_lr")})
    n0 = 6
    try:
        opts["iterations"] = float(n0)
        rops.training_once(h, opts, 0)          # warm-up: MIOpen picks its kernels, the allocator fills
        t1 = rops.training_once(h, opts, 0)
        opts["iterations"] = float(n0 + steps)
        t2 = rops.training_once(h, opts, 0)
        log = rops.log(h)
        peak = torch.cuda.max_memory_allocated(dev)
    finally:
        rops.destroy(h)
        torch.cuda.empty_cache()
    ms = (t2 - t1) / steps * 1e3
    import re
    ema = [float(m) for m in re.findall(r"ema_loss:([-\d.]+)", log)]
    return {"steps": steps, "ms_per_step": round(ms, 3), "iters_per_s": round(1e3 / ms, 3),
            "learning_rates": "0 (stationary, like `value`)" if stationary else "training",
            "method": f"GaussianTrainer::trainingOnce with {n0} and {n0 + steps} iterations after a warm-up call; "
                      f"({round(t2, 4)} s - {round(t1, 4)} s) / {steps}",
            "ema_loss_last": ema[-1] if ema else None, "peak_allocated_MB": int(peak // 2**20),
            "host_code": "the reference's src/gaussian_trainer.cpp, gaussian_renderer.cpp, gaussian_rasterizer.cpp, "
                         "GaussianModel members -- compiled unchanged (oracle/_ref/" + name + "); loss: " +
                         ("this repository's host/include/loss_utils.h in place of the reference's header (fused HIP kernels)" if fused_loss
                          else "the reference's include/loss_utils.h (MIOpen convolutions through autograd)"),
            "kernels": "lib/libcuda_rasterizer.so -> libgsr_hip.so through the reference's own signatures (no raw_params, no fused "
                       "loss / Adam / statistics: the reference contract)"}


def multi_gpu_preflight(torch, dist, backend, dev, world, rank, local_rank):
    """Fail LOUDLY, before any timed work, when the node cannot run one rank per GPU over RCCL: device count, process-group
    initialisation, one tiny all-gather and one tiny all-reduce whose results are checked on every rank (the first contact of
    this program with a multi-GPU node is the driver's scaling run: a hang or a silent wrong answer there costs the round's
    only curve).  Returns a dict for the JSON line."""
    n_dev = torch.cuda.device_count()
    shared = os.environ.get("GSR_BENCH_SHARE_GPU") == "1"
    if world > 1 and not shared and n_dev <= local_rank:
        raise SystemExit(f"bench.py preflight: rank {rank} (local rank {local_rank}) has no GPU: {n_dev} device(s) visible, "
                         f"{world} ranks requested (one rank per GPU)")
    t0 = time.perf_counter()
    try:
        import datetime
        kw = dict(timeout=datetime.timedelta(seconds=180))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)
    except Exception as e:
        raise SystemExit(f"bench.py preflight: init_process_group({backend}) failed on rank {rank}: {e!r} "
                         f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, "
                         f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})")
    cdev = dev if backend == "nccl" else torch.device("cpu")
    mine = torch.full((4,), float(rank + 1), device=cdev)
    gathered = torch.empty(world * 4, device=cdev)
    dist.all_gather_into_tensor(gathered, mine)
    want = torch.arange(1, world + 1, device=cdev, dtype=torch.float32).repeat_interleave(4)
    if not torch.equal(gathered, want):
        raise SystemExit(f"bench.py preflight: all-gather over {backend} returned {gathered.tolist()} on rank {rank}, wanted {want.tolist()}")
    total = mine.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    if not torch.equal(total, torch.full((4,), world * (world + 1) / 2.0, device=cdev)):
        raise SystemExit(f"bench.py preflight: all-reduce over {backend} returned {total.tolist()} on rank {rank}")
    if cdev.type == "cuda":
        torch.cuda.synchronize()
    return {"devices_visible": n_dev, "ranks": world, "backend": backend, "all_gather_ok": True, "all_reduce_ok": True,
            "seconds": round(time.perf_counter() - t0, 2)}


def replica_checksum(torch, dist, tensors, dev, backend):
    """Bit-level agreement of the replicas after the timed steps: every parameter tensor as int32 words summed in int64 (wraps:
    a hash, not a norm) -- MIN and MAX over the ranks must agree."""
    h = torch.stack([t.detach().contiguous().view(torch.int32).to(torch.int64).sum() for t in tensors])
    if backend != "nccl":
        h = h.cpu()
    lo, hi = h.clone(), h.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi)), [int(x) for x in h.cpu()]


def mapper_loop_leg(torch, dev, ops, scene, steps, seed, sh_adam_window, morton_reindex=False, persistent_workspace=True):
    """BASELINE config C5's SHAPE on one GPU (`--mapper-loop`, opt-in): 4 M Gaussians @ 752x480, the fused train step cycling
    through EIGHT keyframes, and the map maintenance of GaussianMapper::trainForOneIteration / run (src/gaussian_mapper.cpp:614-774,
    371-542) on its schedule: increasePcd of 5 k new map points every 10 iterations (a new keyframe's points, :854,955),
    densifyAndPrune every 100, one resetOpacity (iteration 150), one oneUpShDegree (iteration 200; the model starts at degree 2).
    Training learning rates.  One HIP event per iteration: it/s over the whole loop and the cost of each kind of event above the
    median plain step.  (ORB-SLAM3's pose feed is not part of the measured step: the keyframes are the scene's camera arc.)"""
    import math
    cl = scene.make_config("C5", seed=seed, n_views=8)
    cams = cl.cameras
    W, H, P = cams[0].W, cams[0].H, cl.xyz.shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bg = torch.zeros(3, device=dev)
    feats = np.concatenate([cl.features_dc, cl.features_rest], 1)
    h = ops.trainer_create(t(cl.xyz), t(feats), t(cl.opacity), t(cl.scaling), t(cl.rotation), 3, float(cl.extent), bg)
    ops.trainer_set_options(h, {"lazy_sh_adam_window": float(sh_adam_window), "densify": 1.0, "cameras_extent": float(cl.extent),
                                "seed": float(seed), "densify_from_iter": 0.0, "densification_interval": 100.0,
                                "opacity_reset_interval": 150.0, "active_sh_degree": 2.0, "morton_reindex": 1.0 if morton_reindex else 0.0,
                                "persistent_workspace": 1.0 if persistent_workspace else 0.0})
    kfs, gts = [], []
    gen = torch.Generator(device="cpu").manual_seed(4321 + seed)
    for c in cams:
        view, proj, cen = t(c.viewmatrix), t(c.projmatrix), t(c.campos)
        fov = (2 * math.atan(c.tanfovx), 2 * math.atan(c.tanfovy))
        with torch.no_grad():
            img = ops.trainer_render(h, view, proj, cen, fov[0], fov[1], H, W, False, False, True)[0]
        noise = torch.nn.functional.avg_pool2d(torch.rand(3, H, W, generator=gen).unsqueeze(0), 5, 1, 2).squeeze(0).to(dev)
        kfs.append((view, proj, cen, fov))
        gts.append((img.detach() + 0.1 * (noise - 0.5)).clamp_(0.0, 1.0))
    mask = torch.ones(3, H, W, device=dev)
    rng = np.random.default_rng(5 + seed)
    new_pts = t(rng.uniform([-3, -1.5, -3], [3, 1.5, 3], (5000, 3)).astype(np.float32))
    new_cols = torch.rand(5000, 3, generator=torch.Generator().manual_seed(6)).to(dev)

    def step(i):
        view, proj, cen, fov = kfs[i % 8]
        ops.trainer_render_and_backward(h, view, proj, cen, fov[0], fov[1], H, W, gts[i % 8], mask)
        ops.trainer_finish(h)           # statistics, densifyAndPrune / resetOpacity on their schedule, Adam

    for i in range(8):                  # warm-up (allocator, lazy rows): iterations 1..8 of the schedule, untimed
        step(i)
    # the first insertion moves the 4 M Gaussians this leg STARTS with into the arena (two sets of parameters + moments: 8.7 GB of
    # fresh allocations and a copy of everything, ~140 ms once -- a SLAM session's map starts small and grows inside the arena)
    ops.trainer_increase_pcd(h, new_pts[:64] * 1.01, new_cols[:64], 8, False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    kinds = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    points = [int(ops.trainer_params(h)[0].shape[0])]
    for i in range(steps):
        it = 8 + i + 1                  # the trainer's iteration counter of this step
        step(8 + i)
        kind = "plain"
        if it % 100 == 0:
            kind = "densifyAndPrune"
        if it % 150 == 0:
            kind = "resetOpacity" if kind == "plain" else kind + "+resetOpacity"
        if it == 200:
            ops.trainer_one_up_sh_degree(h)
            kind = kind if kind != "plain" else "oneUpShDegree"
        if it % 10 == 0:
            ops.trainer_increase_pcd(h, new_pts, new_cols, it, False)
            kind = kind + "+increasePcd" if kind != "plain" else "increasePcd"
        kinds.append(kind)
        ev[i + 1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
    plain = float(np.median([per[i] for i in range(steps) if kinds[i] == "plain"]))
    events = {}
    for k in sorted(set(kinds) - {"plain"}):
        v = [float(per[i]) for i in range(steps) if kinds[i] == k]
        events[k] = {"count": len(v), "ms_per_step_median": round(float(np.median(v)), 3), "ms_over_a_plain_step_median": round(float(np.median(v)) - plain, 3)}
        events[k]["ms_per_step_max"] = round(max(v), 3)
        if len(v) <= 4:
            events[k]["ms_each"] = [round(x, 3) for x in v]     # (the first use of an ATen kernel loads its code object: a one-off)
    plain_steps = sorted(((float(per[i]), 8 + i + 1) for i in range(steps) if kinds[i] == "plain"), reverse=True)
    # where a plain step of this loop spends its time: 16 more steps (no maintenance falls on them) with the stage events on
    from photo_slam_amd import capi
    lib = capi.load()
    capi.profile_enable(lib, 1)
    stage = {}
    for i in range(16):
        step(8 + steps + i)
        for k, v in capi.profile_read(lib).items():
            if v >= 0:
                stage.setdefault(k, []).append(v)
    capi.profile_enable(lib, 0)
    torch.cuda.synchronize()
    P_end = int(ops.trainer_params(h)[0].shape[0])
    ops.trainer_destroy(h)
    torch.cuda.empty_cache()
    return {"workload": "C5 shape: EuRoC MH_01, 4 M Gaussians @ 752x480, eight keyframes in rotation, full mapper-loop maintenance, one GPU",
            "steps": steps, "iters_per_s": round(steps / el, 3), "ms_per_step_mean": round(el / steps * 1e3, 3),
            "ms_first_to_last_event": round(float(ev[0].elapsed_time(ev[steps])), 3), "ms_sum_of_steps": round(float(per.sum()), 3), "ms_wall": round(el * 1e3, 3),
            "ms_plain_step_median": round(plain, 3), "ms_plain_step_mean": round(float(np.mean([p for p, _ in plain_steps])), 3),
            "slowest_plain_steps": [{"iteration": it, "ms": round(ms, 3)} for ms, it in plain_steps[:10]], "events": events,
            "stage_ms_median_of_a_plain_step": {k: round(float(np.median(v)), 4) for k, v in stage.items()},
            "stage_note": "16 plain steps behind the timed loop with the stage events on: after 300 training-lr steps the synthetic scene has "
                          "inflated (DESIGN.md section 7) -- the instance-bound stages (emit_instances, tile_sort) carry ~3x the instances of step 1", "gaussians_start": P, "gaussians_end": P_end, "persistent_workspace": bool(persistent_workspace),
            "device_memory_MB": {"allocated_peak": int(torch.cuda.max_memory_allocated() // 2**20), "reserved_peak": int(torch.cuda.max_memory_reserved() // 2**20),
                                 "reserved_at_end": int(torch.cuda.memory_reserved() // 2**20),
                                 "allocator_retries": int(torch.cuda.memory_stats().get("num_alloc_retries", 0))},
            "learning_rates": "training", "sh_degree": "2, then 3 from iteration 200 (oneUpShDegree)",
            "schedule": "increasePcd(5 k) every 10 iterations, densifyAndPrune every 100, resetOpacity every 150, oneUpShDegree at 200",
            "note": "one HIP event per iteration; an iteration's time includes the maintenance calls that follow its optimizer step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--points", type=int, default=None, help="override the number of Gaussians (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="20 instead of 100 CPU iterations of C1")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="run only the CPU train-step baseline (no GPU work)")
    ap.add_argument("--raster-only", action="store_true", help="time rasterizer fwd+bwd only (no loss/optimizer)")
    ap.add_argument("--host", default="cpp", choices=["cpp", "py"],
                    help="host layer driving the step: the LibTorch C++ one (photo-slam_amd/host, default) or its Python mirror")
    ap.add_argument("--densify-interval", type=int, default=0,
                    help="run densifyAndPrune every N steps inside the timed region (0 = off; reference: 100); with several "
                         "ranks the host reduces the per-rank statistics over the ranks right before")
    ap.add_argument("--training-lr", action="store_true",
                    help="time the main leg with the training learning rates (drifting synthetic workload) instead of the "
                         "stationary one")
    ap.add_argument("--sh-adam-window", type=int, default=32,
                    help="lazy Adam steps for the SH rows of culled Gaussians, at most this many at a time (gsr_sh_adam_lazy; "
                         "0 = every row steps eagerly at every iteration)")
    ap.add_argument("--scene-order", default="random", choices=["random", "morton"],
                    help="index order of the synthetic Gaussians: 'random' (as generated: i.i.d. positions, so the ~47 %% a view sees are "
                         "a random subset of every cache line of every per-Gaussian array -- the headline) or 'morton' (sorted along a "
                         "Z-order curve: the index coherence of a map that grows keyframe by keyframe)")
    ap.add_argument("--no-fused-geom-adam", action="store_true",
                    help="xyz / opacity / scaling / rotation step in four separate Adam passes instead of inside the backward kernels")
    ap.add_argument("--densify-leg-steps", type=int, default=250,
                    help="steps of the densify_run leg (densifyAndPrune every 100 steps, training learning rates; 0 = skip)")
    ap.add_argument("--insert-every", type=int, default=0,
                    help="densify_run leg: GaussianModel::increasePcd of 5 k new points every N steps (0 = only the five timed calls "
                         "after the leg)")
    ap.add_argument("--no-knn-leg", dest="knn_leg", action="store_false", help="skip the simple-knn leg (100 k and 1 M points)")
    ap.add_argument("--dropin-steps", type=int, default=20,
                    help="steps of the dropin_unfused leg: the reference's own host code on these kernels (0 = skip)")
    ap.add_argument("--dropin-only", action="store_true", help="run only the dropin_unfused leg (profiling)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic scene (SURVEY.md 8d: 0 for reported numbers, 1-4 for variance)")
    ap.add_argument("--mapper-loop", action="store_true",
                    help="run ONLY the C5-shaped mapper-loop leg (4 M @ 752x480, eight keyframes, increasePcd / densify / opacity reset / "
                         "oneUpShDegree on the mapper's schedule) and print its JSON")
    ap.add_argument("--mapper-loop-steps", type=int, default=300)
    ap.add_argument("--exchange-form", default="auto", choices=["auto", "dense", "packed"],
                    help="data-parallel runs, view-factored exchange: every rank sends its whole [P + 1, 3] colour-gradient buffer (dense), "
                         "only the rows its view sees (packed: include/gsr.h gsr_pack_color_view), or whichever a guarded trial of both "
                         "measures faster on this node (auto)")
    ap.add_argument("--median-steps", type=int, default=100, help="steps of the per-step-event leg (protocol.median_*)")
    ap.add_argument("--dump-steps", action="store_true", help="protocol.step_ms: the per-step times of that leg (debugging)")
    ap.add_argument("--no-persistent-workspace", action="store_true",
                    help="TrainStep allocates the rasterizer's three scratch buffers per call, as the reference's resizeFunctional does, "
                         "instead of keeping them across iterations (RasterWorkspace): the A/B handle of the main leg and of --mapper-loop")
    ap.add_argument("--morton-reindex", action="store_true",
                    help="densify_run / mapper-loop legs: densifyAndPrune lays the new set out along a Z-order curve (GaussianModel::morton_reindex_; "
                         "include/gsr.h: gsr_densify_gather_args.morton_scratch) -- the same Gaussians in another row order")
    ap.add_argument("--no-config-legs", dest="config_legs", action="store_false",
                    help="skip the `configs` block (the other single-GPU BASELINE configs -- C2, a C4 view, a C5 view -- each in a child process)")
    ap.add_argument("--no-sq-probe", dest="sq_probe", action="store_false",
                    help="do not measure the blend kernels' VALU issue with a rocprofv3 counter pass (the committed profile is cited instead)")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        import __graft_entry__ as entry
        entry.load_package()
        from photo_slam_amd import scene
        print(json.dumps(cpu_train_step_baseline(scene, args, 0, 0, quick=args.quick_cpu_baseline)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("GSR_BENCH_FORCE_DP") != "1":
        self_launch(args)

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
        preflight = multi_gpu_preflight(torch, dist, backend, dev, world, rank, local_rank)
    lib = capi.load()  # raises if libgsr_hip.so is missing
    if args.mapper_loop:
        sys.path.insert(0, os.path.join(ROOT, "photo-slam_amd", "host"))
        import build_host
        torch.ops.load_library(build_host.build("hip"))
        print(json.dumps({"mapper_loop": mapper_loop_leg(torch, dev, torch.ops.photoslam_amd, scene, args.mapper_loop_steps, args.seed,
                                                         args.sh_adam_window, args.morton_reindex, not args.no_persistent_workspace)}), flush=True)
        return

    cfg = scene.CONFIGS[args.config]
    cl = scene.make_config(args.config, seed=args.seed, n_views=max(world, 1), P=args.points)
    if args.scene_order == "morton":
        # the same cloud, re-indexed along a Z-order curve (10 bits per axis)
        q = ((cl.xyz - cl.xyz.min(0)) / (np.ptp(cl.xyz, axis=0) + 1e-9) * 1023.0).astype(np.uint64)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
        for name in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
            setattr(cl, name, np.ascontiguousarray(getattr(cl, name)[order]))
    cam = cl.cameras[rank % len(cl.cameras)]
    W, H, P = cam.W, cam.H, cl.xyz.shape[0]
    g = GaussianModel.from_cloud(cl, device=dev)
    opt = GaussianOptimizationParams()
    g.trainingSetup(opt)
    kf = GaussianKeyframe.from_camera(cam, dev)
    bg = torch.zeros(3, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank + 1000 * args.seed)
    pipe = GaussianPipelineParams()
    # Ground truth = this view of the initial model + low-pass noise: a converged scene under refinement.
    noise = torch.nn.functional.avg_pool2d(torch.rand(3, H, W, generator=gen).unsqueeze(0), 5, 1, 2).squeeze(0).to(dev)
    with torch.no_grad():
        gt = (GaussianRenderer.render(kf, H, W, g, pipe, bg)[0] + 0.1 * (noise - 0.5)).clamp_(0.0, 1.0)
    mask = torch.ones(3, H, W, device=dev)
    import math
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    if args.dropin_only:
        print(json.dumps({"dropin_unfused": dropin_unfused_leg(torch, dev, cl, kf, fovx, fovy, H, W, gt, max(args.dropin_steps, 1),
                                                                stationary=not args.training_lr),
                          "dropin_loss_header_swapped": dropin_unfused_leg(torch, dev, cl, kf, fovx, fovy, H, W, gt, max(args.dropin_steps, 1),
                                                                            stationary=not args.training_lr, fused_loss=True),
                          "config": {"workload": f"{args.config}: {cfg['note']}", "gaussians": P, "width": W, "height": H}}), flush=True)
        return
    if args.densify_interval:
        opt.densification_interval_, opt.densify_from_iter_ = args.densify_interval, 0
    ts = TrainStep(g, opt, pipe, bg, world_size=world, cameras_extent=cl.extent,
                   densify=bool(args.densify_interval), factored_exchange=factored, lazy_sh_adam_window=args.sh_adam_window,
                   fused_geom_adam=not args.no_fused_geom_adam)

    ops = None
    if args.host == "cpp" and not args.raster_only:
        sys.path.insert(0, os.path.join(ROOT, "photo-slam_amd", "host"))
        import build_host
        torch.ops.load_library(build_host.build("hip"))
        ops = torch.ops.photoslam_amd
        handle = ops.trainer_create(g.xyz_.detach(), g.features_.detach(), g.opacity_.detach(), g.scaling_.detach(),
                                    g.rotation_.detach(), 3, float(cl.extent), bg)
        ops.trainer_set_options(handle, {"lazy_sh_adam_window": float(args.sh_adam_window),
                                         "fused_geom_adam": 0.0 if args.no_fused_geom_adam else 1.0,
                                         "persistent_workspace": 0.0 if args.no_persistent_workspace else 1.0})
        # GSR_BENCH_PY_EXCHANGE=1: the collectives issued from Python around the C++ pieces (the round-2 arrangement, kept for
        # comparison); default: the C++ host drives the exchange itself on the c10d process group (keyframe_batch_exchange.cpp)
        # (GSR_BENCH_CPP_EXCHANGE=1 keeps the C++ exchange on a backend other than RCCL: the two-ranks-on-one-GPU functional check)
        py_exchange = os.environ.get("GSR_BENCH_PY_EXCHANGE") == "1" or (dp and backend != "nccl"
                                                                          and os.environ.get("GSR_BENCH_CPP_EXCHANGE") != "1")
        if dp:
            ops.trainer_set_options(handle, {"fused_sh_adam": 0.0})   # the optimizer follows the gradient exchange
        counts_over = "nothing to exchange (one rank)" if world == 1 else f"{backend}, on the gather stream"
        if dp and not py_exchange:
            ops.trainer_set_process_group(handle, dist.group.WORLD.group_name, factored)
            if world > 1:
                # the packed exchange agrees on its message capacity host-side: one int per rank over gloo (the visible counts are on
                # the host anyway; through RCCL their two pinned copies cost the compute stream +42 us per step: profiles/r04_r).
                # Every rank takes the same route: a gloo group that does not come up on ANY rank leaves the RCCL route on all.
                count_group, ok = None, 1.0
                if os.environ.get("MASTER_ADDR") in ("127.0.0.1", "localhost"):
                    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # (one node: gloo need not resolve the container's hostname)
                try:
                    import datetime
                    count_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
                    probe = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
                    dist.all_gather(probe, torch.full((1,), rank, dtype=torch.int32), group=count_group)
                    ok = 1.0 if [int(p) for p in probe] == list(range(world)) else 0.0
                except Exception as e:                        # noqa: BLE001
                    ok = 0.0
                    print(f"bench.py: no gloo group for the visible counts on rank {rank} ({e!r}); they travel through {backend}", file=sys.stderr)
                flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) > 0:
                    ops.trainer_set_count_group(handle, count_group.group_name)
                    counts_over = "a gloo group next to the gradients' group (host-side)"
        elif dp and factored:
            ops.trainer_set_factored_exchange(handle, True)
        if args.densify_interval:
            ops.trainer_set_options(handle, {"densify": 1.0, "cameras_extent": float(cl.extent), "seed": 0.0,
                                             "densify_from_iter": 0.0, "densification_interval": float(args.densify_interval)})

    def set_lr_scale(s):
        if ops is not None:
            ops.trainer_set_options(handle, {"lr_scale": float(s)})
        if g.optimizer_ is not None:
            g.optimizer_.lr_scale = float(s)

    # The reference reads the loss on the host every iteration (EMA for logging, gaussian_mapper.cpp:701-705).  So does this
    # loop -- one step late: the value is copied to pinned memory behind the step's kernels and read while the NEXT step is
    # already queued, so the stream never drains for a log value.
    loss_pinned = [torch.zeros(1).pin_memory() for _ in range(2)]
    loss_ready = [torch.cuda.Event() for _ in range(2)]
    loss_state = {"n": 0, "ema": 0.0}
    comm_ms = {"gather": [], "reduce": []}

    no_loss_read = os.environ.get("GSR_BENCH_NO_LOSS_READ") == "1"   # (A/B: what does the per-iteration host read of the loss cost?)

    def read_loss_deferred(loss):
        if no_loss_read:
            return
        k = loss_state["n"] & 1
        if loss_state["n"] >= 1:
            loss_ready[k ^ 1].synchronize()                       # step n-1: finished long ago, or being finished now
            loss_state["ema"] = 0.4 * float(loss_pinned[k ^ 1][0]) + 0.6 * loss_state["ema"]
        loss_pinned[k].copy_(loss.detach().reshape(1), non_blocking=True)
        loss_ready[k].record()
        loss_state["n"] += 1

    kf_now = [kf]   # (the changing-views leg swaps the keyframe between steps)

    def one_step():
        kf = kf_now[0]
        if ops is not None and dp and not py_exchange:
            # one call: render + backward, the exchange (RCCL, issued by the C++ host), the optimizer
            loss = ops.trainer_train_one_iteration(handle, kf.world_view_transform_, kf.full_proj_transform_, kf.camera_center_,
                                                   fovx, fovy, H, W, gt, mask)
            read_loss_deferred(loss)
        elif ops is not None:
            loss = ops.trainer_render_and_backward(handle, kf.world_view_transform_, kf.full_proj_transform_,
                                                   kf.camera_center_, fovx, fovy, H, W, gt, mask)
            if dp and factored:
                # view-factored exchange (trainer.ViewFactoredExchange): the colour gradients are gathered in two halves, the
                # other four tensors reduced; the SH gradient of each half is rebuilt from the views and applied while the
                # next collective is on the links
                grads = ops.trainer_grads(handle)
                ex = ViewFactoredExchange(ops.trainer_sh_send_buffer(handle), kf.camera_center_,
                                          [(i, t) for i, t in enumerate(grads) if i != FEATURES_GROUP], world)
                ops.trainer_finish_begin(handle)
                for first, (row0, centres, views) in enumerate(ex.gathered_parts()):
                    # rebuild + Adam in one pass over the rows SOME view lights (lazy rows, --sh-adam-window); reads xyz: before
                    # ITS Adam
                    ops.trainer_features_step_from_views(handle, centres, views, row0, first == 0)
                ops.trainer_features_finish_from_views(handle)   # this step's slice of the rotating catch-up
                for i in ex.order():
                    ex.wait(i)                # stream-side wait: the host keeps queueing (one collective for the four)
                ops.trainer_geom_adam(handle, ex.reduction_.grad_scale())   # xyz / opacity / scaling / rotation: one Adam launch (+ the 1/N)
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

    def timed(steps, read_dominant=False):
        """EXACTLY `steps` steps between barrier + synchronize; max over ranks.  Returns (seconds, dominant-kernel ms list)."""
        dom = []
        barrier()
        t0 = time.perf_counter()
        stamps = []
        for _ in range(steps):
            one_step()
            if read_dominant:
                v = capi.profile_read(lib)["blend_bwd"]      # waits only for events already recorded this step
                if v >= 0:
                    dom.append(v)
            stamps.append(time.perf_counter())
        barrier()
        el = time.perf_counter() - t0
        if os.environ.get("GSR_BENCH_STAMPS"):
            print("STAMPS", [round((b - a) * 1e3, 3) for a, b in zip([t0] + stamps, stamps + [t0 + el])], file=sys.stderr)
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, dom

    stationary = not args.training_lr
    set_lr_scale(0.0 if stationary else 1.0)
    # ---- the form of the view-factored exchange (data-parallel runs driven by the C++ host): a guarded trial of both forms
    # with frozen parameters (learning rates x 0: the replicas cannot drift apart whatever happens), every rank adopting the
    # decision of the slowest one; a packed form that fails on this node (it has only ever run at one rank on hardware) leaves
    # the dense form in place instead of taking the run down
    exchange_form = None
    if dp and factored and ops is not None and not py_exchange:
        exchange_form = {"requested": args.exchange_form}
        def trial(packed, n=12):
            ops.trainer_set_options(handle, {"packed_exchange": 1.0 if packed else 0.0, "lr_scale": 0.0})
            for _ in range(4):
                one_step()
            el, _ = timed(n)
            return el / n * 1e3
        choice = args.exchange_form
        if choice == "auto":
            for _ in range(max(args.sh_adam_window, 8)):     # (both forms are compared in the lazy rows' steady state)
                one_step()
            t_dense = trial(False)
            try:
                t_packed = trial(True)
                ok = 1.0
            except Exception as e:                            # noqa: BLE001 -- any failure of the untested form means "dense"
                t_packed, ok = float("inf"), 0.0
                exchange_form["packed_trial_error"] = repr(e)[:300]
            flag = torch.tensor([ok], device=dev if backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            choice = "packed" if float(flag.item()) > 0 and t_packed < t_dense else "dense"
            exchange_form.update({"dense_ms_per_step": round(t_dense, 4), "packed_ms_per_step": round(t_packed, 4) if ok else None})
        ops.trainer_set_options(handle, {"packed_exchange": 1.0 if choice == "packed" else 0.0, "lr_scale": 0.0 if stationary else 1.0})
        exchange_form["used"] = choice
    # lazy SH Adam (--sh-adam-window): the rotating catch-up of the culled rows reaches its steady state (every flushed row
    # `window` steps behind) after `window` steps -- the steps that the requested warm-up does not cover are run before it,
    # untimed, so that the timed region does a steady state's work per step
    lazy_dp = dp and factored and ops is not None
    priming = max(0, (args.sh_adam_window if (not (dp or args.raster_only) or lazy_dp) else 0) - args.warmup)
    # (the interpreter's garbage collector stays out of the window: a generation-2 collection inside 20 steps is a 2 ms step.  It runs
    # HERE, in front of the warm-up steps -- a collection right in front of the window idles the device for tens of ms, and the first
    # dozen steps behind such a pause run 5-10 % slow: measured with GSR_BENCH_STAMPS=1, profiles/r06_y)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(priming + args.warmup):
        one_step()
    # Timed region: HIP events only around the backward blend, the dominant kernel (gsr_profile_enable(2)): every event
    # record is a ~5 us bubble in the stream, and eleven of them per step cost 2 % of the step they are meant to measure.
    capi.profile_enable(lib, 2)
    capi.host_wait_stats(lib)                       # (reset)
    elapsed, dom_ms = timed(args.steps, read_dominant=True)
    gc.enable()
    # how long the host was blocked in the forward pass's one synchronisation: a host that runs ahead of the device waits there
    # for most of a step; near zero = the device waits for the host (include/gsr.h: gsr_host_wait_stats)
    host_wait_us, host_waits = capi.host_wait_stats(lib)
    capi.profile_enable(lib, 0)

    # ---- per-step times from HIP events on the stream: 20 warm-up + median of >= 100
    n_med = max(args.median_steps, 0)
    step_ms = []
    if n_med:
        for _ in range(max(0, 20 - args.warmup - args.steps)):
            one_step()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_med + 1)]
        barrier()
        marks[0].record()
        for i in range(n_med):
            one_step()
            marks[i + 1].record()
        barrier()
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_med)]

    # ---- stage table: the same (fused) program with events between all stages
    capi.profile_enable(lib, 1)
    stage_ms = {}
    for _ in range(20):
        one_step()
        for k, v in capi.profile_read(lib).items():
            stage_ms.setdefault(k, []).append(v)
    torch.cuda.synchronize()
    # ---- data-parallel runs: what every rank's compute stream WAITED for (HIP events around each Work::wait() in the C++ host:
    # the exposed communication of the step) and every rank's stage table, so that a bad scaling curve can be read from one run
    exchange_wait = per_rank = None
    if dp and ops is not None and not py_exchange:
        ops.trainer_set_options(handle, {"profile_exchange": 1.0})
        waits = []
        for _ in range(20):
            one_step()
            waits.append(ops.trainer_exchange_wait_ms(handle))
        ops.trainer_set_options(handle, {"profile_exchange": 0.0})
        mine = {"rank": rank, "all_gather_wait_ms_median": round(float(np.median([w[0] for w in waits])), 4),
                "all_reduce_wait_ms_median": round(float(np.median([w[1] for w in waits])), 4),
                "segments_ms_median": {name: round(float(np.median([w[k] for w in waits])), 4) for k, name in
                                       ((2, "forward_loss_backward"), (0, "all_gather_wait"), (3, "sh_step_from_views"),
                                        (1, "all_reduce_wait"), (4, "geometry_adam_and_finish")) if len(waits[0]) > k},
                "stage_ms_median": {k: round(float(np.median([m for m in v if m >= 0])), 4) for k, v in stage_ms.items() if any(m >= 0 for m in v)}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        exchange_wait = {"all_gather_wait_ms_max_over_ranks": max(r["all_gather_wait_ms_median"] for r in per_rank),
                         "all_reduce_wait_ms_max_over_ranks": max(r["all_reduce_wait_ms_median"] for r in per_rank),
                         "method": "HIP events on the compute stream in front of and behind every Work::wait() "
                                   "(host/src/keyframe_batch_exchange.cpp: markWait), median of 20 steps per rank"}
    # ---- the rasterizer alone: the same legs with the SH Adam step as a separate pass (not fused into backward), so that
    # "rendered Mpix/s (fwd+bwd)" prices rasterizer work only; reported next to the fused program's figures, never as `value`
    unfused_ms = {}
    if not dp and not args.raster_only:
        if ops is not None:
            ops.trainer_set_options(handle, {"fused_sh_adam": 0.0})
        ts.fused_sh_adam_ = False
        for _ in range(12):
            one_step()
            for k, v in capi.profile_read(lib).items():
                unfused_ms.setdefault(k, []).append(v)
        torch.cuda.synchronize()
        if ops is not None:
            ops.trainer_set_options(handle, {"fused_sh_adam": 1.0})
        ts.fused_sh_adam_ = True
    capi.profile_enable(lib, 0)

    # ---- scene statistics of this rank's view (V, R) for the byte model -- of the stationary state the legs above ran on
    with torch.no_grad():
        img, _, vis, radii = GaussianRenderer.render(kf, H, W, g, pipe, bg)
        V = int(vis.sum().item())
    from photo_slam_amd import rasterize_points as rp  # noqa
    R = rp.RasterizeGaussiansCUDA(bg, g.getXYZ().detach(), torch.empty(0, device=dev), g.getOpacityActivation().detach(),
                                  g.getScalingActivation().detach(), g.getRotationActivation().detach(), 1.0,
                                  torch.empty(0, device=dev), kf.world_view_transform_, kf.full_proj_transform_,
                                  kf.tanfovx_, kf.tanfovy_, H, W, g.getFeatures().detach(), 3, kf.camera_center_, False)[0]

    # ---- K steps that cycle through four keyframes of the scene's camera arc (same stationary parameters): the culled set
    # changes from step to step, so rows of the lazily stepped SH tensor keep becoming visible and catch up in the forward pass
    views_run = None
    if stationary and not args.raster_only and not dp and ops is not None:
        cams4 = scene.make_config(args.config, seed=args.seed, n_views=4, P=args.points).cameras
        kfs4 = [GaussianKeyframe.from_camera(c, dev) for c in cams4]
        n4 = max(args.steps, 40)
        for i in range(8):
            kf_now[0] = kfs4[i % 4]
            one_step()
        barrier()
        t0 = time.perf_counter()
        for i in range(n4):
            kf_now[0] = kfs4[i % 4]
            one_step()
        barrier()
        el4 = time.perf_counter() - t0
        kf_now[0] = kf
        views_run = {"steps": n4, "keyframes": 4, "ms_per_step": round(el4 / n4 * 1e3, 3), "iters_per_s": round(n4 / el4, 3),
                     "note": "four keyframes on the scene's 1 m camera arc taken in turn (the ground-truth image stays the first "
                             "keyframe's: the loss value is meaningless, the work is not)"}

    # ---- the same K steps with the training learning rates (the drifting synthetic workload), for the record
    train_run = None
    replicas_identical = param_hash = None
    if stationary and not args.raster_only:
        set_lr_scale(1.0)
        el2, _ = timed(args.steps)
        if dp and ops is not None:
            # the replicas after steps that MOVED the parameters (the stationary legs cannot tell replicas apart)
            replicas_identical, param_hash = replica_checksum(torch, dist, ops.trainer_params(handle), dev, backend)
        train_run = {"steps": args.steps, "ms_per_step": round(el2 / args.steps * 1e3, 3),
                     "iters_per_s": round(world * args.steps / el2, 3),
                     "note": "training learning rates from the same start: the synthetic scene inflates ~1 %/step (DESIGN.md section 7)"}

    # ---- 100 steps with the training learning rates from a FRESH model (the drift makes "the first K steps" depend on K: this
    # leg fixes K = 100 whatever --steps is), the C++ host's fused program
    train_run_100 = None
    if stationary and not args.raster_only and not dp and ops is not None and args.densify_leg_steps > 0:
        g3 = GaussianModel.from_cloud(cl, device=dev)
        h3 = ops.trainer_create(g3.xyz_.detach(), g3.features_.detach(), g3.opacity_.detach(), g3.scaling_.detach(),
                                g3.rotation_.detach(), 3, float(cl.extent), bg)
        del g3
        ops.trainer_set_options(h3, {"lazy_sh_adam_window": float(args.sh_adam_window),
                                     "fused_geom_adam": 0.0 if args.no_fused_geom_adam else 1.0})
        def step3():
            read_loss_deferred(ops.trainer_render_and_backward(h3, kf.world_view_transform_, kf.full_proj_transform_, kf.camera_center_,
                                                               fovx, fovy, H, W, gt, mask))
            ops.trainer_finish(h3)
        for _ in range(args.warmup):
            step3()
        barrier()
        t0 = time.perf_counter()
        for _ in range(100):
            step3()
        barrier()
        el3 = time.perf_counter() - t0
        ops.trainer_destroy(h3)
        train_run_100 = {"steps": 100, "warmup": args.warmup, "ms_per_step": round(el3 / 100 * 1e3, 3), "iters_per_s": round(100 / el3, 3),
                         "note": "training learning rates, fresh model, 100 timed steps (no densification)"}

    # ---- the reference's own host code on these kernels: what the API-only swap delivers (never `value`)
    dropin_run = dropin_fl_run = None
    if rank == 0 and world == 1 and not dp and not args.raster_only and args.dropin_steps > 0:
        dropin_run = dropin_unfused_leg(torch, dev, cl, kf, fovx, fovy, H, W, gt, args.dropin_steps)
        dropin_fl_run = dropin_unfused_leg(torch, dev, cl, kf, fovx, fovy, H, W, gt, args.dropin_steps, fused_loss=True)

    # ---- BASELINE config C3 as stated ("with densify/prune + simple-knn"): the same program with the training learning rates and
    # densifyAndPrune every 100 steps (the reference's densification_interval_), on a fresh model; every step timed by its own
    # HIP event so that the densifying steps can be read separately.  `value` stays the stationary leg above.
    densify_run = densify_run_morton = None
    densify_legs = []
    if stationary and not args.raster_only and not dp and ops is not None and not args.densify_interval and args.densify_leg_steps > 0:
        # (twice: the reference's row order, then GaussianModel::morton_reindex_ -- the same Gaussians laid out along a Z-order curve by
        # every densifyAndPrune, include/gsr.h: gsr_densify_gather_args.morton_scratch; --morton-reindex runs only the second)
        densify_legs = [True] if args.morton_reindex else [False, True]
    for leg_morton in densify_legs:
        g2 = GaussianModel.from_cloud(cl, device=dev)
        h2 = ops.trainer_create(g2.xyz_.detach(), g2.features_.detach(), g2.opacity_.detach(), g2.scaling_.detach(),
                                g2.rotation_.detach(), 3, float(cl.extent), bg)
        del g2
        interval = 100
        ops.trainer_set_options(h2, {"lazy_sh_adam_window": float(args.sh_adam_window), "densify": 1.0,
                                     "fused_geom_adam": 0.0 if args.no_fused_geom_adam else 1.0,
                                     "cameras_extent": float(cl.extent), "seed": 0.0, "densify_from_iter": 0.0,
                                     "densification_interval": float(interval), "morton_reindex": 1.0 if leg_morton else 0.0})
        n_d = args.densify_leg_steps
        P_before = int(ops.trainer_params(h2)[0].shape[0])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_d + 1)]
        barrier()
        t0 = time.perf_counter()
        ev[0].record()
        new_pts = torch.from_numpy(np.random.default_rng(5).uniform([-3, -1.5, -3], [3, 1.5, 3], (5000, 3)).astype(np.float32)).to(dev)
        new_cols = torch.rand(5000, 3, generator=torch.Generator().manual_seed(6)).to(dev)
        inserted_at = []
        for i in range(n_d):
            loss = ops.trainer_render_and_backward(h2, kf.world_view_transform_, kf.full_proj_transform_, kf.camera_center_, fovx,
                                                   fovy, H, W, gt, mask)
            ops.trainer_finish(h2)
            if args.insert_every and (i + 1) % args.insert_every == 0:
                # GaussianModel::increasePcd (src/gaussian_model.cpp:188-376): 5 k new map points, as a new keyframe brings them
                ops.trainer_increase_pcd(h2, new_pts, new_cols, i + 1, False)
                inserted_at.append(i)
            read_loss_deferred(loss)
            ev[i + 1].record()
        barrier()
        el_d = time.perf_counter() - t0
        per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n_d)])
        calls = [i for i in range(n_d) if (i + 1) % interval == 0]
        plain = np.array([per[i] for i in range(n_d) if i not in calls])
        P_after = int(ops.trainer_params(h2)[0].shape[0])
        last = [int(x) for x in ops.trainer_last_densify(h2)]
        densify_run = {"steps": n_d, "densification_interval": interval, "learning_rates": "training",
                       "ms_per_step": round(el_d / n_d * 1e3, 3), "iters_per_s": round(n_d / el_d, 3), "densify_calls": len(calls),
                       "ms_per_densifying_step": [round(float(per[i]), 3) for i in calls],
                       "ms_median_other_steps": round(float(np.median(plain)), 3),
                       "ms_per_densify_call_over_a_plain_step": [round(float(per[i] - np.median(plain)), 3) for i in calls],
                       "gaussians_before": P_before, "gaussians_after": P_after, "morton_reindex": bool(leg_morton),
                       "last_call": dict(zip(("cloned", "split", "pruned", "points"), last)),
                       "note": "BASELINE config C3 as stated: densifyAndPrune (src/gaussian_model.cpp:716-815) every 100 steps inside the "
                               "timed loop, training learning rates; a densifying step skips its optimizer update as the reference's does; "
                               "the synthetic scene keeps splitting the same high-gradient Gaussians, so the work per step grows"}
        if inserted_at:
            densify_run["insert_every"] = args.insert_every
            densify_run["ms_per_inserting_step"] = [round(float(per[i]), 3) for i in inserted_at]
        # increasePcd alone: 5 k points into the (by now ~2 M) model, five times in a row -- kNN among the new points + the
        # append (in place while the arena has room; the reference re-cats all six tensors and their moments: 2 x 708 B x P)
        ins = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        P0 = int(ops.trainer_params(h2)[0].shape[0])
        torch.cuda.synchronize()
        ins[0].record()
        for k in range(5):
            ops.trainer_increase_pcd(h2, new_pts, new_cols, n_d + k, False)
            ins[k + 1].record()
        torch.cuda.synchronize()
        densify_run["increase_pcd"] = {"points_per_call": 5000, "calls": 5, "gaussians_before": P0,
                                       "gaussians_after": int(ops.trainer_params(h2)[0].shape[0]),
                                       "ms_per_call": [round(ins[k].elapsed_time(ins[k + 1]), 3) for k in range(5)],
                                       "note": "GaussianModel::increasePcd (src/gaussian_model.cpp:188-376): distCUDA2 among the new "
                                               "points + append with zero moments; the first call may grow the arena"}
        ops.trainer_destroy(h2)
        torch.cuda.empty_cache()
        if leg_morton:
            densify_run_morton = densify_run
            densify_run = None if args.morton_reindex else first_densify_run
        else:
            first_densify_run = densify_run

    # ---- simple-knn (distCUDA2, third_party/simple-knn/simple_knn.cu:185-221): the other half of "with densify/prune + simple-knn"
    knn_run = None
    if rank == 0 and not args.raster_only and not dp and args.knn_leg:
        from photo_slam_amd import rasterize_points as rp2
        knn_run = {"bound_model": "120 B per point when box pruning is effective (SURVEY.md 8d): AABB 12 x 2 + Morton 16 + sort 64 + "
                                  "neighbour scan 16", "runs": []}
        rng_k = np.random.default_rng(7)
        for n_pts in (100_000, 1_000_000):
            pts_np = (rng_k.random((n_pts, 3), dtype=np.float32) * np.array([6, 3, 6], np.float32) - np.array([3, 1.5, 3], np.float32))
            pts = torch.from_numpy(pts_np).to(dev)
            for _ in range(2):
                d = rp2.distCUDA2(pts)
            reps = 5
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            e0[0].record()
            for i in range(reps):
                d = rp2.distCUDA2(pts)
                e0[i + 1].record()
            torch.cuda.synchronize()
            ms = float(np.median([e0[i].elapsed_time(e0[i + 1]) for i in range(reps)]))
            entry = {"points": n_pts, "ms": round(ms, 4), "algorithmic_GBps_at_120B_per_point": round(120.0 * n_pts / (ms * 1e-3) / 1e9, 2),
                     "frac_of_hbm_peak": round(120.0 * n_pts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            if not args.no_cpu_baseline:
                from oracle import oracle as orc
                orc.build()
                t0 = time.perf_counter()
                d_cpu = orc.knn(pts_np)
                entry["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                entry["cpu_oracle_threads"] = orc.get_threads()
                entry["bit_identical_to_cpu_oracle"] = bool(np.array_equal(d.cpu().numpy(), d_cpu))
            knn_run["runs"].append(entry)
        knn_run["note"] = ("exact 3-NN search over a three-level box hierarchy of the Morton order (32 / 1024 / 32768 points, csrc/knn.hip): "
                           "compute-bound in the neighbour scan (sub-box tests + candidate distances; per-kernel split: "
                           "profiles/r03_*_knn_kernel_stats_*.csv), far from the 120 B/point stream bound of SURVEY.md 8(d); round 2's "
                           "scan of whole 1024-point boxes took 4.4 ms / 11.2 ms at these sizes")

    T = ((W + 15) // 16) * ((H + 15) // 16)
    tile_bits = int(np.ceil(np.log2(max(T, 2))))
    fused_sh_adam = not dp and not args.raster_only   # both hosts fuse the SH Adam step into backward at one rank
    lazy_window = args.sh_adam_window if fused_sh_adam and args.sh_adam_window >= 2 else 0
    from photo_slam_amd import capi as _capi
    tile_first = bool(_capi.load().gsr_binning_tile_first(0, P, W, H))   # the binning arrangement gsr_forward takes at this size
    ab = algorithmic_bytes(P, V, R, W * H, T, tile_passes=(tile_bits + 7) // 8, fused_sh_adam=fused_sh_adam, lazy_window=lazy_window,
                           fused_geom_adam=fused_sh_adam and not args.no_fused_geom_adam, tile_first=tile_first)
    stages = {}
    for k, ms in stage_ms.items():
        ms = [m for m in ms if m >= 0]
        if not ms:
            continue
        avg = float(np.median(ms))
        stages[k] = dict(ms=round(avg, 4), bytes=int(ab[k]), GBps=round(ab[k] / (avg * 1e-3) / 1e9, 1) if avg > 0 else None)
    raster_ms = sum(s["ms"] for s in stages.values())
    # the dominant KERNEL: every stage is one kernel except the sorts / scans (3-12 launches) and "preprocess_bwd", which with
    # the Adam step of the SH tensor fused in is three kernels (long_run_sums, preprocess_bwd, sh_bwd_rows: the largest of them
    # 0.3 ms) -- those are not candidates
    multi = {"depth_sort", "offset_scan", "tile_sort", "preprocess_bwd"}
    single = [k for k in stages if k not in multi]
    dom = max(single, key=lambda k: stages[k]["ms"]) if single else None
    if dom == "blend_bwd" and dom_ms:
        # the dominant kernel's duration inside the timed region replaces the stage-table value
        avg = float(np.mean(dom_ms))
        stages[dom] = dict(ms=round(avg, 4), bytes=int(ab[dom]), GBps=round(ab[dom] / (avg * 1e-3) / 1e9, 1),
                           timed_region=True)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        lr_note = ("0 (stationary workload: every kernel runs and the Adam moments update, the parameters do not move)"
                   if stationary else "training (GaussianOptimizationParams defaults; the synthetic scene drifts)")
        out = {
            "metric": "train iters/s (render + L1/SSIM loss + backward + Adam), 2M Gaussians @1080p"
            if args.config == "C3" else f"train iters/s, config {args.config}",
            "value": round(world * args.steps / elapsed, 3),
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_steps_before_warmup": priming,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['note']}", "gaussians": P, "width": W, "height": H,
                       "visible": V, "instances": R, "instances_per_visible": round(R / max(V, 1), 2),
                       "keyframes_per_step": world, "sh_degree": 3,
                       "parallelism": "single GPU" if not dp else
                                      (f"dp{world} (one keyframe per GPU; ONE all-gather of 3 floats/Gaussian + the camera centre, issued "
                                       "behind preprocess_bwd on a second stream, + ONE all-reduce (sum) of 11 floats/Gaussian; SH "
                                       "gradient rebuilt and stepped lazily per rank; densification statistics accumulate per rank)") if factored else
                                      f"dp{world} (one keyframe per GPU, all-reduce of 59 floats/Gaussian)",
                       "raster_only": bool(args.raster_only), "densify_interval": args.densify_interval,
                       "learning_rates": lr_note,
                       "cull_empty_tiles": os.environ.get("GSR_CULL_EMPTY_TILES", "0") == "1",   # (include/gsr.h: same image and gradients)
                       "binning": "tile-first" if tile_first else "depth-first",   # (include/gsr.h: GSR_BINNING_*; the same lists either way)
                       "sh_adam_fused_into_backward": fused_sh_adam,
                       "sh_adam_lazy_window": lazy_window, "scene_index_order": args.scene_order,
                       "geometry_adam_fused_into_backward": bool(fused_sh_adam and not args.no_fused_geom_adam),
                       "gaussians_after": int(g.xyz_.shape[0]) if ops is None else int(ops.trainer_params(handle)[0].shape[0]),
                       "host": "libtorch-c++ (photo-slam_amd/host)" if ops is not None else "python mirror"},
            "mpix_per_s": round(world * W * H / (raster_ms * 1e-3) / 1e6, 1) if raster_ms > 0 else None,
            "raster_fwd_bwd_ms": round(raster_ms, 4),
        }
        if unfused_ms:
            # rasterizer stages without optimizer work (SH Adam as a separate pass): medians of 12 further steps
            ab_u = algorithmic_bytes(P, V, R, W * H, T, tile_passes=(tile_bits + 7) // 8, fused_sh_adam=False, tile_first=tile_first)
            u = {k: float(np.median([m for m in ms if m >= 0])) for k, ms in unfused_ms.items() if any(m >= 0 for m in ms)}
            ums = sum(u.values())
            out["rasterizer_only"] = {"fwd_bwd_ms": round(ums, 4), "mpix_per_s": round(world * W * H / (ums * 1e-3) / 1e6, 1),
                                      "hbm_frac": round(sum(ab_u.values()) / (ums * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "stages_ms": {k: round(v, 4) for k, v in u.items()},
                                      "note": "same step with the SH Adam update as a separate pass after backward: the stage "
                                              "events then bracket rasterizer work only (mpix_per_s above includes the fused update)"}
        if step_ms:
            s = np.sort(np.array(step_ms))
            out["protocol"] = {"timed": "wall clock over exactly `steps` steps between barrier + synchronize (value, ms_per_step)",
                               "median_ms_per_step": round(float(np.median(s)), 4), "p10_ms": round(float(s[len(s) // 10]), 4),
                               "p90_ms": round(float(s[(9 * len(s)) // 10]), 4), "median_over_steps": len(s),
                               "warmup_before_median": max(20, args.warmup + args.steps),
                               "median_iters_per_s": round(world * 1e3 / float(np.median(s)), 3),
                               "source": "one HIP event per step on the compute stream of rank 0"}
            if args.dump_steps:
                out["protocol"]["step_ms"] = [round(float(x), 3) for x in step_ms]
        if host_waits:
            out["host"] = {"blocked_in_forward_sync_us_per_step": round(host_wait_us / max(args.steps, 1), 1), "syncs": host_waits,
                           "note": "time the host thread spent blocked in gsr_forward's one synchronisation during the timed steps "
                                   "(gsr_host_wait_stats): large = the host runs ahead of the device; near zero = the device waits for the host"}
        if views_run:
            out["changing_views_run"] = views_run
        if train_run:
            out["training_lr_run"] = train_run
        if train_run_100:
            out["training_lr_run_100"] = train_run_100
        if densify_run:
            out["densify_run"] = densify_run
        if densify_run_morton:
            out["densify_run_morton_reindex"] = densify_run_morton
        if dropin_run:
            out["dropin_unfused"] = dropin_run
        if dropin_fl_run:
            out["dropin_loss_header_swapped"] = dropin_fl_run
        # BASELINE.json's configuration AS STATED (C3 "with densify/prune") and the non-stationary figures, lifted next to `value`
        # (`value` itself is the stationary leg: lr x 0, one fixed view, lazy rows in steady state -- the most favourable one)
        out["stated_config"] = {
            "value_is": "stationary leg (learning rates x 0, one fixed view)",
            "densify_every_100_training_lr_iters_per_s": densify_run["iters_per_s"] if densify_run else None,
            "densify_every_100_training_lr_morton_reindex_iters_per_s (opt-in: GaussianModel::morton_reindex_)":
                densify_run_morton["iters_per_s"] if densify_run_morton else None,
            "training_lr_100_steps_iters_per_s": train_run_100["iters_per_s"] if train_run_100 else None,
            "changing_views_iters_per_s": views_run["iters_per_s"] if views_run else None,
            "reference_host_code_on_these_kernels_iters_per_s": dropin_run.get("iters_per_s") if dropin_run else None,
            "reference_host_code_with_this_loss_header_iters_per_s": dropin_fl_run.get("iters_per_s") if dropin_fl_run else None}
        if knn_run:
            out["knn"] = knn_run
        if dp:
            out["preflight"] = preflight
            out["exposed_communication"] = exchange_wait
            out["per_rank"] = per_rank
            out["replicas_identical"] = replicas_identical
            out["replica_parameter_hash_rank0"] = param_hash
            out["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                           "collectives_issued_by": "python (trainer.py classes)" if (ops is None or py_exchange) else
                                                    "the C++ host (host/src/keyframe_batch_exchange.cpp on c10d::ProcessGroup)",
                           "NCCL_ALGO": os.environ.get("NCCL_ALGO", "default"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "default"),
                           "exchange": "view-factored" if factored else "all-reduce", "exchange_form": exchange_form}
            if exchange_form:
                out["rccl"]["packed_form_visible_counts_over"] = counts_over
        traffic = traffic_src = None
        # HBM bytes per launch from rocprofv3 TCC counters (separate --pmc passes, tools/gpu_pmc.sh), corrected as
        # MI355X_MICROARCH.md prescribes for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Measured for C3 only, on the
        # committed profile of the same build -- NOT in this run: the source file rides along.
        kernel_of = {"blend_bwd": "gsr::blend_bwd_kernel", "blend_fwd": "gsr::blend_fwd_kernel",
                     "preprocess_fwd": "gsr::preprocess_fwd_kernel", "preprocess_bwd": "gsr::preprocess_bwd_kernel"}
        for f in PMC_FILES:
            if dom and args.config == "C3" and args.points is None and dom in kernel_of and os.path.exists(os.path.join(ROOT, f)):
                table = json.load(open(os.path.join(ROOT, f)))
                pm = table.get(kernel_of[dom]) or next((v for k, v in table.items() if k.startswith(kernel_of[dom])), None)   # (template arguments follow the name)
                if pm:
                    traffic = int((2 * pm.get("FETCH_SIZE", 0) + pm.get("WRITE_SIZE", 0)) * 1024)
                    traffic_src = f
                    break
        if dom == "blend_bwd" and fused_sh_adam and not lazy_window and os.environ.get("GSR_SH_ADAM_SIDE_STREAM", "1") != "0":
            # eager mode: while blend_bwd runs, the library's second stream streams the SH rows of ALL culled Gaussians
            # (gsr_backward).  (Lazy mode: the slice kernel runs behind the blend, next to the preprocess_bwd stage, whose
            # algorithmic bytes include it.)
            side_bytes = 1152 * (P - V)
            stages[dom]["concurrent"] = {"kernel": "sh_adam_culled_kernel (side stream, 1152 B per culled Gaussian)",
                                         "bytes": int(side_bytes),
                                         "combined_GBps_if_fully_overlapped": round((ab[dom] + side_bytes) / (stages[dom]["ms"] * 1e-3) / 1e9, 1)}
        if dom:
            a = stages[dom]["GBps"]
            # what the dominant kernel is ACTUALLY bound by: the blend kernels issue VALU instructions at the machine's rate while
            # HBM idles (achieved / peak / frac stay the HBM figures the contract asks for, so that the distance is on record);
            # the utilisation is measured here when the profiler is on the box (one rocprofv3 --pmc pass in a child), else cited
            sq_live = sq_note = None
            if args.sq_probe and world == 1 and not dp and args.points is None:
                sq_live, sq_note = sq_probe(args.config, args.seed)
            bounds = stage_bounds(stages, sq_live=sq_live, sq_live_note=sq_note)
            dom_b = bounds.get(dom, {})
            out["roofline"] = {"kernel": dom, "bound": dom_b.get("bound", "hbm"), "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(a / HBM_PEAK_GBS, 4),
                               "valu_issue_utilisation": dom_b.get("valu_issue_utilisation"),
                               "valu_issue_source": dom_b.get("sq_counters_source"),
                               "traffic": traffic,
                               "traffic_source": traffic_src and f"{traffic_src} (rocprofv3 --pmc of the same build; not measured in this run)",
                               # the rasterizer alone (SH Adam as a separate pass: rasterizer bytes / rasterizer time); the fused
                               # program's figure -- whose backward also carries the optimizer's bytes -- under its own key
                               "raster_fwd_bwd_frac": out.get("rasterizer_only", {}).get("hbm_frac"),
                               "fused_step_stage_bytes_frac": round(sum(ab.values()) / (raster_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "bounds": bounds,
                               "stage_table_source": "median of 20 further steps of the SAME program with HIP events between all "
                                                     "stages (preprocess_bwd carries the fused Adam step of the SH tensor when "
                                                     "config.sh_adam_fused_into_backward); the dominant kernel's entry is its mean "
                                                     "duration inside the timed region",
                               "stages": stages}
        if world == 1 and not dp and args.config_legs and args.config == "C3" and args.points is None and not args.raster_only:
            # the other single-GPU configs of BASELINE.json in the SAME record (C2 = Replica's native resolution; one keyframe of
            # the C4 and C5 batches = the per-rank work of the 8-GPU runs), each with its own roofline entry.  This process's
            # model is released first: a C5 view wants 4 M Gaussians of its own.
            torch.cuda.synchronize()
            out["configs"] = {}
            for c in ("C2", "C4", "C5"):
                out["configs"][c] = config_leg(c, args.seed)
            # the C3 workload itself with the model's rows along a Z-order curve (the order GaussianModel::morton_reindex_ leaves behind
            # every densifyAndPrune): the synthetic cloud's i.i.d. order is the worst case for the per-Gaussian kernels' cache lines
            out["configs"]["C3_rows_in_z_order"] = config_leg("C3", args.seed, steps=args.steps, warmup=args.warmup,
                                                              extra=("--scene-order", "morton", "--no-sq-probe", "--median-steps", "100"))
            out["configs"]["C3_rows_in_z_order"]["note"] = ("same cloud, same view, same instances; rows sorted along a Z-order curve "
                                                            "(--scene-order morton), the main leg's protocol (same steps / warm-up, median of 100 further steps).  Not `value`: that "
                                                            "keeps the cloud as generated")
        if world == 1 and not args.no_cpu_baseline:
            base, kept = cpu_train_step_baseline(scene, args, W, H, quick=args.quick_cpu_baseline, want_inputs=True)
            main = base.get(args.config, base["C1"])
            if ops is not None and "C1" in kept:
                # the SAME sequence on the GPU: C1 cloud, the CPU run's ground truth, training learning rates, as many iterations
                # as the CPU run took -- the fused program's losses next to the reference step's (tests/test_train_sequence_reference.py
                # asserts the parameters as well, with a densification and an opacity reset in the sequence)
                k1 = kept["C1"]
                cl1 = k1["cloud"]
                cam1 = cl1.cameras[0]
                g1 = GaussianModel.from_cloud(cl1, device=dev)
                h1 = ops.trainer_create(g1.xyz_.detach(), g1.features_.detach(), g1.opacity_.detach(), g1.scaling_.detach(),
                                        g1.rotation_.detach(), 3, float(cl1.extent), bg)
                ops.trainer_set_options(h1, {"lazy_sh_adam_window": float(args.sh_adam_window),
                                             "fused_geom_adam": 0.0 if args.no_fused_geom_adam else 1.0})
                kf1 = GaussianKeyframe.from_camera(cam1, dev)
                gt1 = torch.from_numpy(k1["gt"]).to(dev)
                mask1 = torch.ones(3, cam1.H, cam1.W, device=dev)
                import math as _m
                gl = []
                for _ in range(len(k1["losses"])):
                    l1_ = ops.trainer_render_and_backward(h1, kf1.world_view_transform_, kf1.full_proj_transform_, kf1.camera_center_,
                                                          2 * _m.atan(cam1.tanfovx), 2 * _m.atan(cam1.tanfovy), cam1.H, cam1.W, gt1, mask1)
                    ops.trainer_finish(h1)
                    gl.append(l1_)
                gl = [float(x) for x in torch.stack(gl).cpu()]
                ops.trainer_destroy(h1)
                cl_ = np.array(k1["losses"])
                base["C1"]["gpu_fused_step_same_sequence"] = {
                    "iterations": len(gl), "loss_first": round(gl[0], 5), "loss_last": round(gl[-1], 5),
                    "cpu_loss_first": round(float(cl_[0]), 5), "cpu_loss_last": round(float(cl_[-1]), 5),
                    "max_rel_diff_over_all_iterations": float(np.max(np.abs(np.array(gl) - cl_) / cl_)),
                    "note": "C++ host, fused program (lazy SH Adam, geometry Adam in backward) on the MI355X vs the reference's step on the "
                            "host cores: same cloud, same ground truth, same learning rates, every iteration's loss compared"}
            out["cpu_baseline"] = {
                "value": main["iters_per_s"], "unit": "iters/s (full train step)", "cores": main["oracle_threads"],
                "kind": "port",
                "sample": (f"{main['iterations']} measured iterations (+{main['warmup']} warm-up) of the reference's train step "
                           f"(src/gaussian_trainer.cpp:45-133) at {main['config']}: CPU oracle rasterizer (port of the reference "
                           f"kernels, pinned bit for bit to them) behind the autograd Function, LibTorch-CPU activations / loss "
                           f"({main['loss_ops']}) / Adam; median {main['s_per_iteration_median']} s per iteration; C1 in full "
                           f"({base['C1']['iterations']} iterations) under `runs`"),
                "runs": base}
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
