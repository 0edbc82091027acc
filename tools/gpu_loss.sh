#!/bin/bash
# loss kernel times (and the loss parity test) for each flag set
mkdir -p gpurun_out; export TMPDIR=/tmp
for flags in "$@"; do
  echo "=== FLAGS: $flags"
  GSR_EXTRA_FLAGS="$flags" python photo-slam_amd/build.py > gpurun_out/build_exp.log 2>&1 || { tail -20 gpurun_out/build_exp.log; continue; }
  timeout 200 python -m pytest tests/test_train_ops.py -m gpu -q 2>&1 | tail -1
  rm -rf /tmp/kp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/kp.log 2>&1)
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "loss_" in r["Name"]: print(r["Name"].split("(")[0], "avg us", round(float(r["AverageNs"])/1e3, 1))
PY
done
