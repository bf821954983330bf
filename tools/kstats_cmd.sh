#!/bin/bash
# Per-kernel device times (rocprofv3 --kernel-trace --stats) of an arbitrary python tool: tools/kstats_cmd.sh <tag> <script> [args...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/s -o $TAG -- python3 $REPO/"$@" > $OUT/run.log 2>&1 || echo "stats pass failed"
DB=$(find $OUT -name "*_results.db" | head -1)
python3 $REPO/tools/rocpd_summary.py stats $DB > $OUT/kernel_stats.csv 2>> $OUT/run.log
grep -v amdgpu.ids $OUT/run.log | tail -2
head -6 $OUT/kernel_stats.csv | cut -c1-150
