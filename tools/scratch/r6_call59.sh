#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
PGQ_WCC_TRACE=1 timeout 600 python tools/scratch/r6_wcc.py 2>&1 | grep -i "wcc" | tail -20
