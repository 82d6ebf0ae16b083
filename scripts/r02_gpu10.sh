#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python scripts/prof_fused_synth.py 1024 0.17 2>&1 | tail -8
python scripts/prof_fused_synth.py 1024 1.0 2>&1 | tail -8
