#!/bin/bash
# per-kernel average durations (rocprofv3 kernel trace) of the raster-only bench, for each flag set
mkdir -p gpurun_out; export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== FLAGS: $flags"
  GSR_EXTRA_FLAGS="$flags" python photo-slam_amd/build.py > gpurun_out/build_exp.log 2>&1 || { tail -20 gpurun_out/build_exp.log; continue; }
  rm -rf /tmp/kp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --raster-only --no-cpu-baseline > /tmp/kp.log 2>&1)
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "gsr::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{r["Name"].split("(")[0].replace("void ",""):44s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
done
