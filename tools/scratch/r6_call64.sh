#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 600 python tools/scratch/r6_repro.py 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-400
