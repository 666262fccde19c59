#!/bin/bash
# Build libnadm.so (gfx950 only) next to this script.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $FLAGS -c nadm_genotype_passes.hip -o nadm_genotype_passes.o "$@"
hipcc $FLAGS -c nadm_small_kernels.hip -o nadm_small_kernels.o "$@"
hipcc $FLAGS -c nadm_step.hip -o nadm_step.o "$@"
hipcc $FLAGS -x hip -c nadm_gmm.cpp -o nadm_gmm.o "$@"          # host code only (decoder-init mixture fit)
hipcc $FLAGS -c nadm_gmm_dev.hip -o nadm_gmm_dev.o "$@"         # the same fit with the sums over the samples on the device
hipcc $FLAGS -c nadm_calib.hip -o nadm_calib.o "$@"             # measurement helper: the box fingerprint of bench.py (not on the training path)
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libnadm.so -o libnadm.so nadm_genotype_passes.o nadm_small_kernels.o nadm_step.o nadm_gmm.o nadm_gmm_dev.o nadm_calib.o -lpthread -ldl
echo "built $(pwd)/libnadm.so"
# the TEST build: the same sources with -DNADM_TEST_HOOKS (nadm_test_force_slices / nadm_test_force_generic_mlp exist only here).  The tests that
# need a hook re-run themselves in a child process against it (tests/conftest.py: in_hook_build); the shipping library has none.
hipcc $FLAGS -DNADM_TEST_HOOKS -c nadm_genotype_passes.hip -o nadm_genotype_passes_th.o "$@"
hipcc $FLAGS -DNADM_TEST_HOOKS -c nadm_small_kernels.hip -o nadm_small_kernels_th.o "$@"
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libnadm.so -o libnadm_testhooks.so nadm_genotype_passes_th.o nadm_small_kernels_th.o nadm_step.o nadm_gmm.o nadm_gmm_dev.o nadm_calib.o -lpthread -ldl
echo "built $(pwd)/libnadm_testhooks.so"
