#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c16
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "concurrent or zero_copy or golden or ball" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for ba in 3 1000; do echo "block_above $ba"; PGQ_BLOCK_ABOVE=$ba timeout 300 python tools/chunk_throughput.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in sorted(d):
    if 'rows_per_s' in k: print('  ', k, '%.1f M rows/s'%(d[k]/1e6), '%.3f ms/chunk'%d[k.replace('rows_per_s','ms_per_chunk')])
"; done
