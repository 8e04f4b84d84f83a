#!/bin/bash
# kernel stats of the cached decode loop (tools/decode_bench.py) for one batch size
set -o pipefail
tag=${1:-profd}; n=${2:-32}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dstats -- python3 $root/tools/decode_bench.py --episodes $((n/2)) --reps 2 > $out/dstats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/dstats.log; exit 1; }
cd $root
f=$(find $out/dstats -name "*kernel_stats.csv" | head -1); cp $f $out/decode_kernel_stats.csv
t=$(find $out/dstats -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $out/decode_by_launch.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "decode" in n or "advance" in n or "msda_fwd" in n:
        k = (n.split("(")[1 if n.startswith("(") else 0][:40] if False else n[:60], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?"))
        a = agg[k]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:60s} grid {k[1]:>8s} lds {k[2]:>7s} calls {c:6d} avg {ns / c / 1e3:7.2f} us total {ns / 1e6:8.2f} ms")
PY
cat $out/decode_by_launch.txt
find $out/dstats -name "*.csv" -size +4M -delete
tail -3 $out/dstats.log
