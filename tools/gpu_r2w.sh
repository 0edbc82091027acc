#!/bin/bash
# round-2 session W: the changing-views leg of bench.py (four keyframes in turn), lazy window 32 against eager, twice
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
run() {
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --median-steps 0 "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  fixed view ms/step', d['ms_per_step'], ' changing views', d['changing_views_run']['ms_per_step'], ' training lr', d['training_lr_run']['ms_per_step'])
"
}
for rep in 1 2; do echo "lazy 32"; run; echo "eager"; run --sh-adam-window 0; done
