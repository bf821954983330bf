#!/bin/bash
# Quick SQ counter passes of the headline sweep (separate rocprofv3 --pmc runs, kernel trace only): instruction mix,
# LDS activity / bank conflicts, VALU busy and wait cycles.  Output: gpurun_out/pmc_<tag>/summary.txt
set -o pipefail
TAG=${1:-q}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  echo "== pmc pass $C" >> $OUT/log.txt
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/$N -o $TAG -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu --no-other-paths --no-full-select > /dev/null 2>> $OUT/log.txt || echo "pmc pass $C failed" >> $OUT/log.txt
done
python3 $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*_results.db") > $OUT/summary.txt 2>> $OUT/log.txt
grep -A14 "cs2_kernel" $OUT/summary.txt | head -40
