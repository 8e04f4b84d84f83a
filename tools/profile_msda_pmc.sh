#!/bin/bash
# LDS counters of the MSDA backward kernels on the encoder shape (rocprofv3 --pmc only, no trace domains).
set -o pipefail
tag=${1:-msda_pmc}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES \
  --output-format csv -d $out/pmc -- python3 $root/tools/msda_bench.py > $out/pmc.log 2>&1 || { echo "pmc pass failed"; tail -5 $out/pmc.log; exit 1; }
cd $root
python3 tools/pmc_summary.py $out/pmc msda > $out/summary.txt
find $out/pmc -name "*.csv" -size +4M -delete
cat $out/summary.txt
