#!/bin/bash
# On the GPU box: the general-graph cheapest_path_length workload under every build_variants/libpgq_hip_relax_*.so (built
# here with tools/build_variants.sh file=pgq_cheapest "relax_<name>:<flags>" ...; they travel with the snapshot unless
# build_variants/ is listed in .gpurunignore) next to the shipped library: 512 pairs with one batch in flight (the kernel
# itself) and 4096 pairs as shipped.  One line per library.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/sweep_relax; mkdir -p $O
for lib in "" build_variants/libpgq_hip_relax_*.so; do
	[ -n "$lib" ] && [ ! -f "$lib" ] && continue
	tag=${lib:-shipped}
	for cfg in "512 1" "4096 6"; do
		set -- $cfg
		PGQ_HIP_LIB=${lib:+$R/$lib} PGQ_RELAX_STREAMS=$2 timeout 300 python bench.py --workload snb_cheapest --steps 1 --warmup 1 --no-cpu-baseline --pairs-per-gpu $1 > $O/b.json 2> $O/b.err
		python -c "
import json; d=json.load(open('$O/b.json')); print('$tag', 'pairs=$1 streams=$2', round(d['ms_per_step'],1), 'ms', round(d['pairs_per_s']), 'pairs/s', d['roofline_by_kernel'].get('relax'))"
	done
done
