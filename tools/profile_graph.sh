#!/bin/bash
# kernel stats of the hipGraph replay alone (bench.py --graph): sum of kernel durations against the wall time of a step
set -o pipefail
tag=${1:-profg}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/gstats -- python3 $root/bench.py --graph --steps 8 --warmup 3 --no_cpu_baseline --no_decode --no_roofline > $out/gstats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/gstats.log; exit 1; }
cd $root
f=$(find $out/gstats -name "*kernel_stats.csv" | head -1); cp $f $out/graph_kernel_stats.csv
t=$(find $out/gstats -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 4 replayed steps: find the adamw kernels as step boundaries
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
b = idx[-9], idx[-1]          # 2 adamw launches per step -> 4 steps
seg = rows[b[0] + 1: b[1] + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = sorted((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) for i in range(len(seg) - 1))
print(f"4 replayed steps: {len(seg)} kernels, wall {wall / 4e6:.3f} ms/step, kernel busy {busy / 4e6:.3f} ms/step, "
      f"median gap {gaps[len(gaps) // 2] / 1e3:.2f} us, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, overlapped pairs {sum(1 for g in gaps if g < 0)}")
PY
find $out/gstats -name "*.csv" -size +4M -delete
grep '^{' $out/gstats.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line under the profiler:', d['ms_per_step'], 'ms/step')"
