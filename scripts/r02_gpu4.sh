#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_4
mkdir -p $OUT
cd $REPO
python scripts/pcie_probe.py 2>&1 | tail -2
HSA_ENABLE_SDMA=0 python scripts/pcie_probe.py 2>&1 | tail -1
nproc; cat /sys/fs/cgroup/cpu.max; lspci 2>/dev/null | grep -i -E "amd|instinct|display" | head -5
