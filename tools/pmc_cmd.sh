#!/bin/bash
# SQ counter passes of an arbitrary python tool: tools/pmc_cmd.sh <tag> <script> [args...]  -> gpurun_out/pmc_<tag>/summary.txt
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/$N -o $TAG -- python3 $REPO/"$@" > /dev/null 2>> $OUT/log.txt || echo "pmc pass $C failed" >> $OUT/log.txt
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- python3 $REPO/"$@" > $OUT/run.txt 2>> $OUT/log.txt
python3 $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*_results.db" -not -path "*stats*") > $OUT/summary.txt 2>> $OUT/log.txt
python3 $REPO/tools/rocpd_summary.py stats $OUT/stats/${TAG}_results.db > $OUT/stats.csv 2>> $OUT/log.txt
cat $OUT/run.txt | tail -2
