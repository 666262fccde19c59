#!/bin/bash
# SQ counters of the pass-2 kernel (two --pmc passes; counters only, no tracing domains besides kernel dispatch).
# usage: tools/pmc_decode.sh [bench flags]   -> gpurun_out/pmc_sq_*.txt
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  (cd /tmp && rocprofv3 --pmc $set -d /tmp/pmc_$i -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/pmc_$i.log 2>&1)
  db=$(find /tmp/pmc_$i -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db ${PMC_KERNEL:-decode_bce} > gpurun_out/pmc_sq_$i.txt 2>&1 || tail -5 /tmp/pmc_$i.log
  cat gpurun_out/pmc_sq_$i.txt
done
