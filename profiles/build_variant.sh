#!/bin/bash
# Build a variant of the HIP library for A/B runs: profiles/build_variant.sh <name> [-DFOO=1 ...]  -> tfrec_amd/ab/<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
mkdir -p $R/tfrec_amd/ab /tmp/ab_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DTFREC_AMD_EXPERIMENTS"
for f in frontend chains chains2 capi; do
	/opt/rocm/bin/hipcc $FLAGS "$@" -c ${SRC:-$R}/tfrec_amd/csrc/$f.hip -o /tmp/ab_$name/$f.o &
done
wait || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tfrec_amd/ab/$name.so /tmp/ab_$name/*.o
echo built $R/tfrec_amd/ab/$name.so
