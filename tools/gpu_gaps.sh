#!/bin/bash
# GPU idle gaps between consecutive kernels of the full train step (rocprofv3 kernel trace timestamps)
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py > gpurun_out/build.log 2>&1
rm -rf /tmp/kt; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline > /tmp/kt.log 2>&1)
tail -1 /tmp/kt.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows))
# keep the last 8 steps (from one forward's first kernel to the next forward's first kernel)
starts = [i for i, e in enumerate(ev) if "preprocess_fwd" in e[2]]
ev = ev[starts[6]:starts[14] + 1]   # 4 warm-up + 12 timed steps (+ bench's trailing forward): steps 7..14
gaps = collections.defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    gaps[(n0[:34], n1[:34])].append(max(0, s1 - e0))
    busy += e0 - s0
span = ev[-1][1] - ev[0][0]
print(f"8 steps: span {span/8e6:.3f} ms/step, kernels busy {busy/8e6:.3f} ms/step, idle {(span-busy)/8e3:.1f} us/step ({100*(span-busy)/span:.1f} %)")
tot = sorted(((sum(v), len(v), k) for k, v in gaps.items()), reverse=True)[:14]
for s, n, (a, b) in tot:
    print(f"  {a:34s} -> {b:34s} n {n:3d} avg gap {s/n/1e3:7.1f} us  per step {s/8e3:7.1f} us")
PY
