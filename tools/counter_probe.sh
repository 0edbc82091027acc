#!/bin/bash
# tools/counter_probe.sh CONFIG OUTDIR "COUNTERS ..." [bench args]: one rocprofv3 --pmc pass (counters only, with --kernel-trace) over a short
# bench.py run; prints the per-kernel means of the blend kernels.  (gpurun: tools/gpu.sh TAG probe:...)
cfg=$1; out=$2; counters=$3; shift 3
rm -rf /tmp/cp_pass; mkdir -p $out
(cd /tmp && TMPDIR=/tmp timeout 500 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/cp_pass -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --median-steps 0 --densify-leg-steps 0 --no-knn-leg --dropin-steps 0 --no-config-legs --no-sq-probe "$@" > /tmp/cp_pass.log 2>&1)
python - "$out/counters_${cfg}_$(echo $counters | tr ' ' '_' | cut -c1-60).json" <<'PY'
import csv, glob, collections, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("/tmp/cp_pass/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            if "gsr::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in out.items():
    if "blend" in k or "tile_depth" in k:
        print(k, {c: round(x, 1) for c, x in v.items()})
if not out:
    print(open("/tmp/cp_pass.log").read()[-1500:])
PY
