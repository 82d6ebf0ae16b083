#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python scripts/prof_fused.py 256 32 sha256xor.gcf 2>&1 | tail -8
