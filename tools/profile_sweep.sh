#!/bin/bash
# PMC passes of the sweep kernels on the 10 M-point workload (tools/exp_sweep.py), through gpurun.
#   tools/profile_sweep.sh <tag> [N] ; outputs gpurun_out/pmc_<tag>/summary.txt
set -o pipefail
TAG=${1:-x}
N=${2:-1e7}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  NAME=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C -d $OUT/p_$NAME -o $TAG -- python3 $REPO/tools/exp_sweep.py $N 4 > $OUT/run_$NAME.log 2>&1 || echo "pass $C failed" >> $OUT/log.txt
done
python3 $REPO/tools/rocpd_summary.py pmc $(find $OUT -name "*_results.db") > $OUT/summary_all.txt 2>> $OUT/log.txt
grep -A14 -E "cs2_kernel|brick_kernel<1" $OUT/summary_all.txt > $OUT/summary.txt
cat $OUT/summary.txt
