#!/bin/bash
# round-2 session G: fused SH Adam with the culled rows on a second stream during the backward blend -- parity and A/B timing
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_train_ops.py tests/test_cpp_host.py -m gpu -q -x > gpurun_out/test_gpu_g.log 2>&1; tail -3 gpurun_out/test_gpu_g.log | cut -c1-300
for side in 1 0; do
  echo "=== GSR_SH_ADAM_SIDE_STREAM=$side"
  GSR_SH_ADAM_SIDE_STREAM=$side timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'train_lr', d['training_lr_run']['ms_per_step'])
print(' '.join(f'{k}={v[\"ms\"]:.4f}' for k, v in d['roofline']['stages'].items()))
"
done
rm -rf /tmp/kp; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --median-steps 0 --training-lr > /tmp/kp.log 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
