#!/bin/bash
# per-kernel averages inside the FULL train step for each flag set (kernel-name filter in $KFILTER, default all gsr kernels)
mkdir -p gpurun_out; export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== FLAGS: $flags"
  GSR_EXTRA_FLAGS="$flags" python photo-slam_amd/build.py > gpurun_out/build_exp.log 2>&1 || { tail -20 gpurun_out/build_exp.log; continue; }
  rm -rf /tmp/kp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/kp.log 2>&1)
  tail -1 /tmp/kp.log | cut -c1-160
  python - <<'PY'
import csv, glob, os
f = glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True)
flt = os.environ.get("KFILTER", "gsr::")
tot = 0
for r in csv.DictReader(open(f[0])):
    tot += float(r["TotalDurationNs"])
    if flt in r["Name"]: print(f'  {r["Name"].split("(")[0].replace("void ",""):42s} avg us {float(r["AverageNs"])/1e3:8.1f}  per-step us {float(r["TotalDurationNs"])/13/1e3:8.1f}')
print("  total kernel ms/step", tot/13/1e6)
PY
done
