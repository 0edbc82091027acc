#!/bin/bash
# round-2 session U: the kernels of ONE train step in launch order with their durations and the gaps between them (which tiny
# launches are left?), memory copies / fills included
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf /tmp/kt; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 4 --median-steps 0 --no-cpu-baseline > /tmp/kt.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one step in the middle of the timed region: from one preprocess_fwd to the next
idx = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
a, b = idx[50], idx[51]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:80]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
    prev_end = max(prev_end, e)
print("step", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
PY
