#!/bin/bash
# Profiles of one round (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. separate --pmc passes (HBM traffic, SQ activity) of a short bench run
# Outputs land in gpurun_out/prof_<tag>/; the summaries are copied to profiles/ by hand.
set -o pipefail
TAG=${1:-r01f}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== stats pass" | tee $OUT/log.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-e2e > $OUT/bench_under_rocprof.json 2>> $OUT/log.txt || echo "stats pass failed" | tee -a $OUT/log.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  echo "== pmc pass $C" | tee -a $OUT/log.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o $TAG -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu --no-other-paths > /dev/null 2>> $OUT/log.txt || echo "pmc pass $C failed" | tee -a $OUT/log.txt
done
find $OUT -name "*_kernel_stats.csv" | head -3 | tee -a $OUT/log.txt
python3 $REPO/tools/rocpd_summary.py pmc $(find $OUT -path "*pmc_*" -name "*_results.db") > $OUT/pmc_summary.txt 2>> $OUT/log.txt
tail -1 $OUT/bench_under_rocprof.json | cut -c1-300
