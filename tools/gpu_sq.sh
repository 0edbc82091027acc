#!/bin/bash
# SQ counters per gsr kernel (VALU / SALU / LDS instruction counts, cycles).  PMC_MODE=full: the fused train step (loss, optimizer
# kernels) instead of the rasterizer alone -> gpurun_out/sq_counters_full.json
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
rm -rf /tmp/sq; (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $( [ "$PMC_MODE" = full ] || echo --raster-only ) --no-cpu-baseline --median-steps 0 > /tmp/sq.log 2>&1)
tail -2 /tmp/sq.log | cut -c1-160
python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "gsr::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    # VALU pipe utilisation: cycles the VALU executes an instruction / (busy cycles x 4 SIMDs per CU are already summed by the counter)
    if d.get("SQ_BUSY_CYCLES"):
        d["valu_active_per_busy"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_BUSY_CYCLES"]
    if d.get("SQ_INSTS_VALU"):
        d["valu_cycles_per_inst"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_INSTS_VALU"]
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_INSTS_VALU"):
        d["kernel_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0     # the counter sums the 8 XCDs
        # SIMD cycles available per wave64 VALU instruction issued: to be read against the issue cost of the kernel's
        # instruction mix (tools/isa_cost.py with profiles/r02_a_valu_rate.json: ~3.9 cycles for blend_fwd, ~4.5 for blend_bwd);
        # there is no single "cycles per VALU instruction" on gfx950 (2.5 ... 8.2 by class)
        d["simd_cycles_per_valu_inst"] = 1024 * d["kernel_cycles"] / d["SQ_INSTS_VALU"]
import os
json.dump(out, open("gpurun_out/sq_counters_full.json" if os.environ.get("PMC_MODE") == "full" else "gpurun_out/sq_counters.json", "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:14]:
    print(k, {c: f"{v:.3g}" for c, v in d.items()})
PY
