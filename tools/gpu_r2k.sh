#!/bin/bash
# round-2 session K: per-kernel durations early vs late in a run with densification every 40 steps (kernel trace, no stats)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -rf /tmp/kt; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --densify-interval 40 --median-steps 420 --no-cpu-baseline > /tmp/kt.log 2>&1)
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(f"{'kernel':44s} {'n':>6s} {'first 30 (us)':>14s} {'last 30 (us)':>14s}")
for k, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    if len(v) < 100 or "gsr::" not in k: continue
    a = [d for _, d in v[40:70]]; b = [d for _, d in v[-230:-200]]   # (the tail of the run is the training-lr / stage legs)
    print(f"{k:44s} {len(v):6d} {sum(a)/len(a):14.1f} {sum(b)/len(b):14.1f}")
PY
