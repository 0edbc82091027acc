#!/bin/bash
# per-launch durations, in order, of selected kernels inside the full train step (is the variance periodic?)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
rm -rf /tmp/kt; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --no-cpu-baseline $BENCH_ARGS > /tmp/kt.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name in ("blend_bwd_kernel", "blend_fwd_kernel", "preprocess_fwd_kernel", "sh_bwd_rows"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    print(name, " ".join(f"{x:.0f}" for x in d[4:44]))
PY
