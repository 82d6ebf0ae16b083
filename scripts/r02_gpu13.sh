#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
GC_TRACE=1 python scripts/bench_stream.py 3000000 2>&1 | grep "gc trace" | tail -24
