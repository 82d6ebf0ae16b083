#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r02_fuzz
timeout 1500 python tests/ext_fuzz.py ${1:-120} > gpurun_out/r02_fuzz/fuzz.log 2>&1; echo "fuzz rc=$?"
grep -E "done|FAIL" gpurun_out/r02_fuzz/fuzz.log | head -30
