#!/bin/bash
# rocprofv3 evidence for one round: kernel stats of the bench command, then the two PMC passes for the GEMM family's HBM
# traffic (separate passes, no trace domains beside --kernel-trace).  Usage: bash tools/profile_round.sh <dir> [<profiles tag, e.g. r03>] ; output
# under gpurun_out/<tag>/, summaries to be copied into profiles/.
set -o pipefail
tag=${1:-prof}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="--steps 4 --warmup 2 --no_cpu_baseline --no_decode --no_roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $B > $out/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $out/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py $B > $out/fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py $B > $out/write.log 2>&1 || { echo "write pass failed"; exit 1; }
cd $root
python3 tools/gemm_traffic.py $out/fetch $out/write $out/gemm_traffic.json
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
# keep the merge-back small: the per-dispatch traces are large
find $out/stats $out/fetch $out/write -name "*.csv" -size +4M -delete
# the bench line of the same tree on the same box (with the roofline leg; profiles/<tag>_gemm_traffic.json must already hold this
# tree's traffic for `traffic_stale: false`, so the script copies it first -- on the GPU box only: after the call, copy
# gpurun_out/<dir>/{gemm_traffic.json,kernel_stats.csv,bench_line.json} into profiles/ by hand and run tools/round_summary.py)
if [ -n "$2" ]; then cp $out/gemm_traffic.json profiles/$2_gemm_traffic.json; fi
python3 bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err || { echo "bench failed"; tail -5 $out/bench.err; exit 1; }
head -12 $out/kernel_stats.csv | cut -c1-200
