#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c17
mkdir -p $O
cd $R
for ba in 3 1000; do echo "block_above $ba"; PGQ_BLOCK_ABOVE=$ba timeout 300 python tools/chunk_throughput.py > $O/ct_$ba.json 2>$O/ct_$ba.err; python -c "
import json,sys
d=json.load(open('$O/ct_$ba.json'))
for k in d: print('  ', k, '%.1f M rows/s'%(d[k]['rows_per_s']/1e6), '%.3f ms/chunk'%d[k]['ms_per_chunk'])
"; tail -2 $O/ct_$ba.err; done
