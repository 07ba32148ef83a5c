#!/bin/bash
# Compile-time variants of one kernel file for tuning sweeps: build_variants/libpgq_hip_<tag>.so (selected with PGQ_HIP_LIB).
# usage: tools/build_variants.sh [file=pgq_meet] "tag:-DPGQ_MEET3_DEPTH=4" "tag2:-DX=1 -DY=2" ...
#        tools/build_variants.sh file=pgq_cheapest "segcond:-DPGQ_RELAX_SEGCOND=1" "w8:-DPGQ_RELAX_WAVES=8"
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/duckpgq-extension_amd/csrc
mkdir -p $R/build_variants
F=pgq_meet
for spec in "$@"; do
	case "$spec" in file=*) F=${spec#file=}; continue;; esac
	tag=${spec%%:*}; flags=${spec#*:}
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C -DNDEBUG $flags -c -o $R/build_variants/${F}_$tag.o $C/$F.hip
	objs=""
	for o in pgq_runtime pgq_msbfs pgq_lanes pgq_meet pgq_analytics pgq_cheapest; do
		if [ $o = $F ]; then objs="$objs $R/build_variants/${F}_$tag.o"; else objs="$objs $C/$o.o"; fi
	done
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libpgq_hip_$tag.so $objs
	rm -f $R/build_variants/${F}_$tag.o
	echo built $tag
done
