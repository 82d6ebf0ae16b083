#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_2
mkdir -p $OUT
cd $REPO
timeout 300 tools/issue_rate_ubench > $OUT/issue_rate.txt 2>&1; echo "ubench rc=$?"
cat $OUT/issue_rate.txt
rocm-smi --showclocks 2>/dev/null | head -20
