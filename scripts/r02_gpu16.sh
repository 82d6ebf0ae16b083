#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
GC_TRACE=1 python scripts/bench_stream.py 2000000 2>&1 | grep "eval:" | tail -8
