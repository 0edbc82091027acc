#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
rm -rf /tmp/sq; (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --raster-only --no-cpu-baseline > /tmp/sq.log 2>&1)
tail -2 /tmp/sq.log | cut -c1-160
python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if r["Kernel_Name"].startswith("gsr::blend") or r["Kernel_Name"].startswith("gsr::preprocess") or r["Kernel_Name"].startswith("gsr::radix_scatter"):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: f"{sum(v)/len(v):.3g}" for c, v in d.items()})
PY
