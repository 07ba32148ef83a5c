#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
PGQ_MEET=0 bash tools/prof_quick.sh r6c49/prof --workload rmat22_cross --no-first-call | head -16
