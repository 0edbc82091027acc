#!/bin/bash
# HBM traffic per launch of every gsr kernel from the TCC counters (separate passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# PMC_MODE=full: the fused train step (loss, SH / geometry Adam inside backward, lazy rows) instead of the rasterizer alone;
# the result then goes to gpurun_out/pmc_traffic_full.json
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $( [ "$PMC_MODE" = full ] || echo --raster-only ) --no-cpu-baseline --median-steps 0 > /tmp/pmc_$c.log 2>&1)
  tail -2 /tmp/pmc_$c.log | cut -c1-200
  find /tmp/pmc_$c -name "*counter_collection.csv" | head -2
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counter file for", c); continue
    acc = collections.defaultdict(list)
    with open(fs[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == c and "gsr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
        out[k]["launches_" + c] = len(v)
import os
json.dump(out, open("gpurun_out/pmc_traffic_full.json" if os.environ.get("PMC_MODE") == "full" else "gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    f, w = v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)
    print(f"{k:32s} FETCH_SIZE {f:12.1f} KB  WRITE_SIZE {w:12.1f} KB   corrected HBM bytes/launch = {(2*f + w)*1024/1e6:9.1f} MB")
PY
