#!/bin/bash
# Build a variant of libnadm.so into tools/abl/<name>.so:  tools/build_variant.sh <name> [-DFLAG=..]...
# (A/B runs: NADM_LIB=tools/abl/<name>.so python bench.py ...; tools/abl_run.sh runs bench.py against every variant.)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
src=$R/neural-admixture_amd/csrc
out=$R/tools/abl
mkdir -p $out /tmp/abl_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $FLAGS -c $src/nadm_genotype_passes.hip -o /tmp/abl_$name/a.o "$@" &
hipcc $FLAGS -c $src/nadm_small_kernels.hip -o /tmp/abl_$name/b.o "$@" &
hipcc $FLAGS -c $src/nadm_step.hip -o /tmp/abl_$name/c.o "$@" &
hipcc $FLAGS -x hip -c $src/nadm_gmm.cpp -o /tmp/abl_$name/d.o "$@" &
hipcc $FLAGS -c $src/nadm_gmm_dev.hip -o /tmp/abl_$name/e.o "$@" &
hipcc $FLAGS -c $src/nadm_calib.hip -o /tmp/abl_$name/f.o "$@" &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libnadm.so -o $out/$name.so /tmp/abl_$name/a.o /tmp/abl_$name/b.o /tmp/abl_$name/c.o /tmp/abl_$name/d.o /tmp/abl_$name/e.o /tmp/abl_$name/f.o -lpthread -ldl
echo "built $out/$name.so"
