#!/bin/bash
# One parameterised GPU session script (replaces the per-session tools/gpu_r2?.sh of round 2).
#   gpurun --timeout 1500 -- 'tools/gpu.sh TAG step [step ...]'
# TAG names the output set gpurun_out/TAG/ (e.g. r03_a); copy what is to be judged into profiles/ as TAG_<file>.
# Steps (run in the order given; each prints a one-line digest):
#   build        python __graft_entry__.py
#   tests        the whole -m gpu suite            tests:<expr>  only tests matching -k <expr>
#   smoke        __graft_entry__.smoke()
#   driver       the driver's command `bench.py --gpus 1 --steps 20 --warmup 5` (with the CPU baseline, timed)
#   bench        `bench.py --no-cpu-baseline` (100 steps + median of 100)      bench:C2 | bench:C4 | bench:C5  other configs
#   big:<points> the C3 view with <points> Gaussians (default 16 M): the path at 8x the stated model size
#   raster       `bench.py --raster-only --no-cpu-baseline`
#   kstats       rocprofv3 --kernel-trace --stats of the driver's command (no CPU baseline) -> kernel_stats_bench_full_C3.csv   kstats:C2  another config
#   knnstats     rocprofv3 --kernel-trace --stats of distCUDA2 at 100 k and 1 M points -> knn_kernel_stats_<points>.csv
#   pmc          TCC traffic per launch, rasterizer only (two --pmc passes)      pmc:full  the fused train step   pmc:full:C2  another config
#   sq           SQ counters (VALU / SALU / LDS issue, busy cycles), fused step  sq:raster  rasterizer only   sq:full:C2  another config
#   dp           the data-parallel program at ONE rank over RCCL (GSR_BENCH_FORCE_DP=1): C3 and a C4 view, view-factored
#   dp:allreduce the same with the plain all-reduce      dp:py  view-factored with the collectives issued from Python
#   dpstats      rocprofv3 --kernel-trace --stats of the data-parallel program at one rank -> kernel_stats_dp_path_1rank_C3.csv
#   share2       N ranks on the box's one GPU over gloo, C++ exchange (share2 | share2:packed | share2:dense | share2:4): functional check of the multi-rank glue
#   mapper       bench.py --mapper-loop: the C5-shaped mapper loop on one GPU (4 M @ 752x480, eight keyframes, map maintenance) -> mapper_loop_C5.json
#   dropin       bench.py --dropin-only: the reference's own host code (oracle/_ref/libref_host_hip.so) on these kernels, 20 steps at C3
#   dropinstats  rocprofv3 --kernel-trace --stats of that leg -> kernel_stats_dropin_unfused_C3.csv
#   seeds        benchq for scene seeds 0..4 -> seed_spread_C3.json (seeds 1-4 for variance)
#   py:<file>    python tools/<file> (an experiment script), output -> <file>.log
#   env:VAR=VAL  export VAR=VAL for the steps behind it (A/B runs; bench outputs get a _VAR_VAL suffix)
#   abenv:VAR=A,B[:n[:config]] alternate the quick bench with VAR=A and VAR=B (an environment switch of the library), n pairs on one box
#   ab:<dir>[:n] alternate the quick bench of the tree in <dir> (a built worktree of an older commit) and of this tree, n pairs on one box
TAG=${1:-r03_x}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}

kernel_stats() {  # $1 = output csv, rest = command
  local out=$1; shift
  rm -rf /tmp/kp; (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- "$@" > /tmp/kp.log 2>&1)
  local f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp $f $out; head -14 $out | cut -c1-130; else echo "no kernel_stats.csv"; tail -5 /tmp/kp.log; fi
  # idle time of the device between consecutive kernels of a step, from the same trace (${out%.csv}_gaps.json): the host
  # wait in the middle of the forward pass (the instance count sizes the binning buffer) and everything else
  local t=$(find /tmp/kp -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python - "$t" "${out%.csv}_gaps.json" <<'PY'
import csv, json, sys, statistics as st
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step = from one preprocess_fwd to the next
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[2]]
steps = []
short = lambda n: n.replace("gsr::", "").replace("void ", "").split("(")[0][:40]
pairs = []   # per step: {(kernel before the gap, kernel behind it): idle ns}
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    span = ks[-1][1] - ks[0][0]
    busy_end, idle, sync_gap, last, where = ks[0][1], 0, None, ks[0][2], {}
    for i in range(1, len(ks)):
        gap = ks[i][0] - busy_end
        if gap > 0:
            idle += gap
            k = (short(last), short(ks[i][2]))
            where[k] = where.get(k, 0) + gap
            if "emit_instances" in ks[i][2]:
                sync_gap = gap
        if ks[i][1] >= busy_end:
            busy_end, last = ks[i][1], ks[i][2]
    steps.append((span, idle, sync_gap, len(ks)))
    pairs.append(where)
steps = steps[len(steps) // 3:]   # (the warm-up third of the run is dropped)
pairs = pairs[len(pairs) - len(steps):]
if steps:
    med = lambda k: st.median(s[k] for s in steps if s[k] is not None) / 1e3
    out = {"steps": len(steps), "kernels_per_step_median": st.median(s[3] for s in steps),
           "step_span_us_median": round(med(0), 1), "device_idle_us_per_step_median": round(med(1), 1),
           "idle_before_emit_instances_us_median (the mid-forward host wait)": round(med(2), 1),
           "note": "from rocprofv3 --kernel-trace timestamps of the same run; concurrent kernels (second stream) count as busy"}
    keys = set(k for w in pairs for k in w)
    top = sorted(((st.median(w.get(k, 0) for w in pairs) / 1e3, k) for k in keys), reverse=True)[:12]
    out["largest_gaps_us_per_step_median"] = [{"after": k[0], "before": k[1], "us": round(v, 1)} for v, k in top if v > 0]
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out))
PY
}

pmc_pass() {  # $1 = mode (raster|full), $2 = config (default C3) -> $OUT/pmc_traffic_<mode>[_<config>].json
  local mode=$1 cfg=${2:-C3}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $ROOT/bench.py --config $cfg --steps 3 --warmup 1 \
        $( [ "$mode" = full ] || echo --raster-only ) --no-cpu-baseline --median-steps 0 --densify-leg-steps 0 --no-knn-leg --dropin-steps 0 > /tmp/pmc_$c.log 2>&1)
  done
  python - "$OUT/pmc_traffic_$mode$( [ $cfg = C3 ] || echo _$cfg ).json" <<'PY'
import csv, glob, collections, json, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter file for", c); continue
    acc = collections.defaultdict(list)
    with open(fs[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == c and "gsr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
        out[k]["launches_" + c] = len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    f, w = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print(f"{k:36s} FETCH {f:11.1f} KB  WRITE {w:11.1f} KB  HBM bytes/launch (2F+W) = {(2*f + w)*1024/1e6:8.1f} MB")
PY
}

sq_pass() {  # $1 = mode (raster|full), $2 = config (default C3) -> $OUT/sq_counters_<mode>[_<config>].json
  local mode=$1 cfg=${2:-C3}
  local sets=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE")
  local i=0
  for set in "${sets[@]}"; do
    rm -rf /tmp/sq_$i
    (cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$i -o p -- python $ROOT/bench.py --config $cfg --steps 3 --warmup 1 \
        $( [ "$mode" = full ] || echo --raster-only ) --no-cpu-baseline --median-steps 0 --densify-leg-steps 0 --no-knn-leg --dropin-steps 0 > /tmp/sq_$i.log 2>&1)
    i=$((i+1))
  done
  python - "$OUT/sq_counters_$mode$( [ $cfg = C3 ] || echo _$cfg ).json" <<'PY'
import csv, glob, collections, json, sys
out = {}
for d in glob.glob("/tmp/sq_[0-9]"):
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(fn) as f:
            for r in csv.DictReader(f):
                if "gsr::" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:12]:
    print(f"{k:36s} " + "  ".join(f"{c.replace('SQ_', '')}={x:.3g}" for c, x in sorted(v.items())))
PY
}

for step in "$@"; do
  arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  echo "=== $step"
  case ${step%%:*} in
    build)  python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }; tail -1 $OUT/build.log ;;
    tests)  if [ -n "$arg" ]; then timeout 1200 python -m pytest tests -q -m gpu -k "$arg" -s --durations=8 > $OUT/test_gpu.log 2>&1; else timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $OUT/test_gpu.log 2>&1; fi
            grep -v Warning $OUT/test_gpu.log | tail -18 | cut -c1-300 ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -1 ;;
    driver) ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_cmd_C3.log 2>&1
            grep "^{\"metric\"" $OUT/bench_driver_cmd_C3.log > $OUT/bench_driver_cmd_C3.json; grep real $OUT/bench_driver_cmd_C3.log
            python - $OUT/bench_driver_cmd_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "raster frac", d["roofline"]["raster_fwd_bwd_frac"])
print("stages", {k: v["ms"] for k, v in d["roofline"]["stages"].items()})
print("stated_config", json.dumps(d.get("stated_config")))
print("dropin_unfused", json.dumps({k: v for k, v in d.get("dropin_unfused", {}).items() if k in ("ms_per_step", "iters_per_s", "method", "skipped", "peak_allocated_MB")}))
for k in ("densify_run", "knn", "changing_views_run", "training_lr_run", "training_lr_run_100"):
    if k in d: print(k, json.dumps(d[k])[:600])
cb = d.get("cpu_baseline", {})
print("cpu", cb.get("value"), json.dumps(cb.get("runs", {}).get("C1", {}).get("gpu_fused_step_same_sequence")))
PY
            ;;
    bench)  cfg=${arg:-C3}; timeout 500 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --dropin-steps 0 $( [ $cfg = C3 ] || echo --densify-leg-steps 0 ) > $OUT/bench_full_$cfg$SUF.json 2>$OUT/bench_err.log
            python -c "
import json; d=json.load(open('$OUT/bench_full_$cfg$SUF.json')); print('$cfg$SUF', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'raster', d.get('rasterizer_only', {}).get('fwd_bwd_ms'), {k: v['ms'] for k, v in d['roofline']['stages'].items()})" ;;
    benchq) cfg=${arg:-C3}; timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/benchq_$cfg$SUF.json 2>$OUT/bench_err.log   # quick A/B form: no densify leg
            python -c "
import json; d=json.load(open('$OUT/benchq_$cfg$SUF.json')); print('$cfg$SUF', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'host blocked us/step', d.get('host', {}).get('blocked_in_forward_sync_us_per_step'), {k: v['ms'] for k, v in d['roofline']['stages'].items()})" ;;
    big)    # big:<points>: the C3 view with <points> Gaussians (default 16 M) -- does the path hold at 8x / 16x the stated model size?
            n=${arg:-16000000}; timeout 900 python bench.py --points $n --steps 20 --warmup 5 --median-steps 20 --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/bench_C3_${n}_points.json 2>$OUT/big_err.log
            python -c "
import json; d=json.load(open('$OUT/bench_C3_${n}_points.json')); print('C3 view,', d['config']['gaussians'], 'Gaussians:', d['ms_per_step'], 'ms', d['value'], 'it/s', 'visible', d['config']['visible'], 'instances', d['config']['instances'], {k: v['ms'] for k, v in d['roofline']['stages'].items()})" || tail -5 $OUT/big_err.log ;;
    raster) timeout 400 python bench.py --raster-only --no-cpu-baseline --no-knn-leg > $OUT/bench_raster_only_C3.json 2>>$OUT/bench_err.log; cut -c1-200 $OUT/bench_raster_only_C3.json ;;
    kstats) cfg=${arg:-C3}; kernel_stats $OUT/kernel_stats_bench_full_$cfg.csv python $ROOT/bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --densify-leg-steps 0 --no-knn-leg --dropin-steps 0 ;;
    knnstats) for n in 100000 1000000; do kernel_stats $OUT/knn_kernel_stats_$n.csv python $ROOT/tools/knn_probe.py $n; done ;;
    pmc)    IFS=: read m c <<< "$arg"; pmc_pass ${m:-raster} ${c:-C3} ;;   # pmc | pmc:full | pmc:full:C2
    sq)     IFS=: read m c <<< "$arg"; sq_pass ${m:-full} ${c:-C3} ;;       # sq | sq:raster | sq:full:C2
    dp)     # dp | dp:allreduce | dp:py (view-factored, collectives issued from Python as in round 2) | dp:late (GSR_EARLY_GATHER=0)
            ex=factored; pyx=0; form=auto; [ "$arg" = allreduce ] && ex=allreduce; [ "$arg" = py ] && pyx=1
            [ "$arg" = packed ] && form=packed; [ "$arg" = dense ] && form=dense   # dp:packed | dp:dense: the form of the view-factored exchange (default: the guarded trial)
            [ "$arg" = late ] && export GSR_EARLY_GATHER=0 || unset GSR_EARLY_GATHER
            for cfg in ${DP_CONFIGS:-C3 C4}; do
              GSR_BENCH_FORCE_DP=1 GSR_BENCH_EXCHANGE=$ex GSR_BENCH_PY_EXCHANGE=$pyx timeout 400 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 --exchange-form $form > $OUT/dp_$cfg.log 2>$OUT/dp_err.log
              f=$OUT/bench_${cfg}_dp_path_1rank_rccl_${arg:-factored}$SUF.json
              grep '^{"metric"' $OUT/dp_$cfg.log > $f   # (the RCCL banner precedes the JSON line)
              python -c "
import json; d=json.load(open('$f')); print('$cfg dp 1 rank ${arg:-factored}', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], d['rccl']['collectives_issued_by'][:12], d['rccl'].get('exchange_form'), d.get('host', {}).get('blocked_in_forward_sync_us_per_step'), (d.get('per_rank') or [{}])[0].get('segments_ms_median'))" || tail -5 $OUT/dp_err.log
            done ;;
    dpstats) GSR_BENCH_FORCE_DP=1 kernel_stats $OUT/kernel_stats_dp_path_1rank_C3${arg:+_$arg}$SUF.csv python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --densify-leg-steps 0 --no-knn-leg --median-steps 0 --dropin-steps 0 --exchange-form ${arg:-dense} ;;   # dpstats | dpstats:packed
    share2) # share2 | share2:packed | share2:dense | share2:4 -- TWO (or N) ranks on the ONE GPU of the box over gloo, the exchange driven
            # by the C++ host: a functional check of bench.py's multi-rank glue (trial of the exchange forms, per-rank tables, replica
            # checksum) -- not a rate: the ranks share the device and gloo moves the bytes through the host
            n=2; form=auto; case "$arg" in packed|dense) form=$arg ;; [0-9]*) n=$arg ;; esac
            GSR_BENCH_SHARE_GPU=1 GSR_BENCH_BACKEND=gloo GSR_BENCH_CPP_EXCHANGE=1 GSR_EXCHANGE_ALLOW_GLOO_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
              --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps 10 --warmup 3 --config C2 --no-cpu-baseline --no-knn-leg \
              --densify-leg-steps 0 --dropin-steps 0 --median-steps 20 --exchange-form $form > $OUT/share_${n}_$form.log 2>$OUT/share_err.log
            f=$OUT/bench_C2_${n}ranks_one_gpu_gloo_cpp_exchange_$form.json
            grep '^{"metric"' $OUT/share_${n}_$form.log > $f
            python -c "
import json; d=json.load(open('$f')); print('C2 $n ranks one GPU gloo', d['n_gpus'], d['ms_per_step'], d['rccl'], 'replicas_identical', d.get('replicas_identical'), d.get('exposed_communication'), 'per_rank', len(d.get('per_rank') or []))" || tail -20 $OUT/share_err.log ;;
    mapper) timeout 900 python bench.py --mapper-loop $( [ "$arg" = morton ] && echo --morton-reindex ) $( [ "$arg" = percall ] && echo --no-persistent-workspace ) > $OUT/mapper_loop_C5$( [ -n "$arg" ] && echo _$arg ).json 2>$OUT/mapper_err.log; cut -c1-700 $OUT/mapper_loop_C5$( [ -n "$arg" ] && echo _$arg ).json; tail -3 $OUT/mapper_err.log ;;   # mapper | mapper:morton | mapper:percall (the rasterizer's scratch buffers allocated per call, as the reference)
    densify) # the stated-config leg alone (250 steps, densify every 100, training learning rates): the reference's row order, then morton_reindex
            timeout 600 python bench.py --steps 5 --warmup 2 --median-steps 0 --no-cpu-baseline --no-knn-leg --dropin-steps 0 --no-config-legs --no-sq-probe > $OUT/densify_run_C3.json 2>>$OUT/bench_err.log
            python -c "
import json; j=json.load(open('$OUT/densify_run_C3.json'))
for k in ('densify_run', 'densify_run_morton_reindex'):
    d=j[k]; print(k, d['iters_per_s'], {q: d[q] for q in d if q in ('ms_per_step', 'ms_median_other_steps', 'ms_per_densifying_step', 'gaussians_after')})" ;;
    drv20)  # drv20[:n]: the driver's protocol (20 timed steps behind 5 warm-up steps) n times in a row on one box: the spread of `value`
            for i in $(seq 1 ${arg:-5}); do
              timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 --no-config-legs --no-sq-probe > $OUT/drv20_$i.json 2>>$OUT/bench_err.log
              python -c "
import json; d=json.load(open('$OUT/drv20_$i.json')); print('run $i: timed', d['ms_per_step'], 'ms  value', d['value'], ' median of 100 further steps', d['protocol']['median_ms_per_step'], 'p90', d['protocol']['p90_ms'])"
            done ;;
    zorder) # zorder[:n]: the driver's protocol at C3 with the cloud as generated and with its rows along a Z-order curve (--scene-order morton), alternating
            for i in $(seq 1 ${arg:-3}); do for o in random morton; do
              timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --scene-order $o --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 --no-config-legs --no-sq-probe > $OUT/zorder_${o}_$i.json 2>>$OUT/bench_err.log
              python -c "
import json; d=json.load(open('$OUT/zorder_${o}_$i.json')); print('run $i $o: timed', d['ms_per_step'], 'ms  value', d['value'], ' median of 100 further steps', d['protocol']['median_ms_per_step'], {k: v['ms'] for k, v in d['roofline']['stages'].items() if k in ('preprocess_fwd', 'blend_fwd', 'blend_bwd', 'preprocess_bwd')})"
            done; done ;;
    dropin) timeout 600 python bench.py --dropin-only > $OUT/dropin_unfused_C3$SUF.json 2>$OUT/dropin_err.log; cut -c1-700 $OUT/dropin_unfused_C3$SUF.json; tail -3 $OUT/dropin_err.log ;;
    dropinstats) kernel_stats $OUT/kernel_stats_dropin_unfused_C3.csv python $ROOT/bench.py --dropin-only --dropin-steps 10 ;;
    seeds)  for sd in 0 1 2 3 4; do timeout 300 python bench.py --seed $sd --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/benchq_C3_seed$sd.json 2>>$OUT/bench_err.log; done
            python - $OUT <<'PY'
import json, sys, statistics as st
out = {"runs": []}
for sd in range(5):
    d = json.load(open(f"{sys.argv[1]}/benchq_C3_seed{sd}.json"))
    out["runs"].append({"seed": sd, "iters_per_s": d["value"], "ms_per_step": d["ms_per_step"], "median_ms_per_step": d["protocol"]["median_ms_per_step"],
                        "visible": d["config"]["visible"], "instances": d["config"]["instances"],
                        "raster_fwd_bwd_ms": d.get("rasterizer_only", {}).get("fwd_bwd_ms"), "blend_bwd_ms": d["roofline"]["stages"]["blend_bwd"]["ms"]})
v = [r["iters_per_s"] for r in out["runs"]]
out["iters_per_s_mean"], out["iters_per_s_stdev"], out["iters_per_s_min"], out["iters_per_s_max"] = round(st.mean(v), 2), round(st.stdev(v), 2), min(v), max(v)
out["note"] = "bench.py --seed S (scene.make_config seed and the ground-truth noise): one box, one session; seed 0 is the reported number"
json.dump(out, open(f"{sys.argv[1]}/seed_spread_C3.json", "w"), indent=1)
print(json.dumps(out)[:900])
PY
            ;;
    ab)     # ab:<dir>[:pairs[:config]]: alternate `bench.py` (quick form) of the tree in <dir> (e.g. a git worktree of the commit before a change,
            # built in this container: it travels with the snapshot) and of this tree on ONE box -> ab_<dir>.json (ms per step, per run)
            IFS=: read d n cfg <<< "$arg"; n=${n:-3}; cfg=${cfg:-C3}
            for i in $(seq 1 $n); do
              (cd $ROOT/$d && timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $ROOT/$OUT/ab_${d}_old_$i.json 2>>$ROOT/$OUT/bench_err.log)
              timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/ab_${d}_new_$i.json 2>>$OUT/bench_err.log
            done
            python - $OUT $d $n $cfg <<'PY'
import json, sys, statistics as st
out, d, n, cfg = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {"config": cfg}
for side in ("old", "new"):
    runs = [json.load(open(f"{out}/ab_{d}_{side}_{i}.json")) for i in range(1, n + 1)]
    res[side] = {"ms_per_step": [r["ms_per_step"] for r in runs], "median_ms_per_step": [r["protocol"]["median_ms_per_step"] for r in runs],
                 "stages_ms_mean": {k: round(st.mean(r["roofline"]["stages"][k]["ms"] for r in runs), 4) for k in runs[0]["roofline"]["stages"]}}
    res[side]["mean_of_medians"] = round(st.mean(res[side]["median_ms_per_step"]), 4)
res["delta_ms (new - old, mean of medians)"] = round(res["new"]["mean_of_medians"] - res["old"]["mean_of_medians"], 4)
res["note"] = f"alternating runs (old, new) x {n} on one box; old = the tree in {d}/"
json.dump(res, open(f"{out}/ab_{d}_{cfg}.json", "w"), indent=1)
print(json.dumps(res))
PY
            ;;
    abenv)  # abenv:VAR=A,B[:pairs]: alternate the quick bench with VAR=A and VAR=B (an environment switch of the library) on ONE box -> abenv_VAR.json
            IFS=: read spec n cfg <<< "$arg"; n=${n:-3}; cfg=${cfg:-C3}   # abenv:VAR=A,B[:pairs[:config]]
            var=${spec%%=*}; vals=${spec#*=}; va=${vals%%,*}; vb=${vals##*,}
            for i in $(seq 1 $n); do
              env $var=$va timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/abenv_${var}_a_$i.json 2>>$OUT/bench_err.log
              env $var=$vb timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-knn-leg --densify-leg-steps 0 --dropin-steps 0 > $OUT/abenv_${var}_b_$i.json 2>>$OUT/bench_err.log
            done
            python - $OUT $var $va $vb $n $cfg <<'PY'
import json, sys, statistics as st
out, var, va, vb, n, cfg = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
res = {"config": cfg}
for side, val in (("a", va), ("b", vb)):
    runs = [json.load(open(f"{out}/abenv_{var}_{side}_{i}.json")) for i in range(1, n + 1)]
    res[f"{var}={val}"] = {"ms_per_step": [r["ms_per_step"] for r in runs], "median_ms_per_step": [r["protocol"]["median_ms_per_step"] for r in runs],
                           "stages_ms_mean": {k: round(st.mean(r["roofline"]["stages"][k]["ms"] for r in runs), 4) for k in runs[0]["roofline"]["stages"]},
                           "mean_of_medians": round(st.mean(r["protocol"]["median_ms_per_step"] for r in runs), 4)}
res["delta_ms (second - first, mean of medians)"] = round(res[f"{var}={vb}"]["mean_of_medians"] - res[f"{var}={va}"]["mean_of_medians"], 4)
res["note"] = f"alternating runs x {n} on one box"
json.dump(res, open(f"{out}/abenv_{var}_{va}_{vb}_{cfg}.json".replace("-", "m"), "w"), indent=1)
print(json.dumps(res))
PY
            ;;
    env)    export "$arg"; SUF="${SUF}_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_')"; echo "exported $arg" ;;   # env:VAR=VALUE for the steps behind it (A/B runs)
    py)     timeout 900 python tools/$arg > $OUT/${arg%.py}.log 2>&1; tail -30 $OUT/${arg%.py}.log | cut -c1-300 ;;
    *)      echo "unknown step $step" ;;
  esac
done
