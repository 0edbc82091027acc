#!/bin/bash
# round-2 session S: the Adam steps of xyz / opacity / scaling / rotation inside the backward kernels -- parity, then the step
# with and without (interleaved), per-kernel times
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_lazy_sh_adam.py tests/test_cpp_host.py tests/test_train_step.py -x -q -m gpu 2>&1 | tail -3
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['roofline']['stages']
print('  ms/step', d['ms_per_step'], 'median', d['protocol']['median_ms_per_step'], 'blend_bwd', s['blend_bwd']['ms'], 'preprocess_bwd', s['preprocess_bwd']['ms'])
"
}
for rep in 1 2; do echo "fused geometry Adam"; run; echo "four separate passes"; run --no-fused-geom-adam; done
cd /tmp; rm -rf /tmp/prof_s; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --median-steps 0 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_s/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(' ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3, 1), 'us')
PY
