#!/bin/bash
# tools/exp_variant_build.sh NAME "-DSK_X=1 ..." -- a variant of the library with other compile-time constants of the scan kernel,
# as build_exp/libsk_NAME.so (git-ignored; travels to the GPU box).  Use: SK_LIBRARY=$PWD/build_exp/libsk_NAME.so python3 tools/exp_grid.py ...
# Build container, after `make -C strainer2_amd/csrc EXPERIMENTS=1` has left the other objects in place.
set -e
NAME=$1; DEFS=$2
cd "$(dirname "$0")/../strainer2_amd/csrc"
mkdir -p ../../build_exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSK_EXPERIMENTS -Wno-unused-value -Wno-unused-result -Wno-unused-function $DEFS -c sk_device.hip -o ../../build_exp/dev_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build_exp/libsk_$NAME.so ../../build_exp/dev_$NAME.o sk_filter.o sk_cover.o sk_host.o sk_host_sd.o sk_host_filter.o sk_host_cov.o -lz -ldl -lpthread
echo built build_exp/libsk_$NAME.so
