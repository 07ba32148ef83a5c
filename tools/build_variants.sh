#!/bin/bash
# Compile-time variants of pgq_meet.hip for tuning sweeps: build_variants/libpgq_hip_<tag>.so (selected with PGQ_HIP_LIB).
# usage: tools/build_variants.sh "tag:-DPGQ_MEET3_DEPTH=4" "tag2:-DX=1 -DY=2" ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/duckpgq-extension_amd/csrc
mkdir -p $R/build_variants
for spec in "$@"; do
	tag=${spec%%:*}; flags=${spec#*:}
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C -DNDEBUG $flags -c -o $R/build_variants/pgq_meet_$tag.o $C/pgq_meet.hip
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libpgq_hip_$tag.so $C/pgq_runtime.o $C/pgq_msbfs.o $C/pgq_lanes.o $R/build_variants/pgq_meet_$tag.o $C/pgq_analytics.o $C/pgq_cheapest.o
	rm -f $R/build_variants/pgq_meet_$tag.o
	echo built $tag
done
