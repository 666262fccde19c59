#!/bin/bash
# UTCL1 (per-CU translation cache) counters of the genotype passes at two resident-matrix sizes: pass 3 is the kernel whose
# duration grows with the size of the resident matrix (52 us at 20k rows, 68 us at 100k).  usage: tools/pmc_tlb.sh
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for rows in 20000 100000; do
  i=0
  for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/pmc_t$i
    (cd /tmp && rocprofv3 --pmc $set -d /tmp/pmc_t$i -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --rows $rows > /tmp/pmc_t$i.log 2>&1)
    db=$(find /tmp/pmc_t$i -name "*.db" | head -1)
    echo "## rows=$rows"
    python $R/tools/pmc_summary.py $db encode_ 2>&1 || tail -5 /tmp/pmc_t$i.log
    python $R/tools/pmc_summary.py $db decode_bce 2>&1 | head -8
  done
done > gpurun_out/pmc_tlb.txt 2>&1
cat gpurun_out/pmc_tlb.txt
