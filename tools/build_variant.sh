#!/bin/bash
# A/B and ablation builds of the screening kernel next to the product library:
#     tools/build_variant.sh NAME [-DSCREEN_VARIANT=n] [-DSCREEN_ABL=n] [more -D flags]  ->  haf_grasping_amd/libhafgrasp_vNAME.so
# The switches are not part of csrc/screen.hip: tools/screen_experiments.patch puts them into a temporary copy first
# (SCREEN_VARIANT bit 0: no B hand-over across the column blocks, bit 1: B fragments two k-steps ahead; SCREEN_ABL, timing only,
# results wrong: 2 no epilogue VALU, 3 no in-loop LDS-DMA, 4 no tile barrier, 5 no B-fragment reads).  Everything else comes from the
# regular build.  Time the variants on ONE GPU box with   HAF_LIB=haf_grasping_amd/libhafgrasp_vNAME.so python tools/time_svm_stage.py
set -e
cd "$(dirname "$0")/../haf_grasping_amd"
N=$1; shift
T=$(mktemp -d)
cp csrc/screen.hip $T/screen.hip
patch -s $T/screen.hip ../tools/screen_experiments.patch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-inline-asm -fno-slp-vectorize \
    -Icsrc "$@" -c $T/screen.hip -o $T/screen.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC csrc/kernels.o $T/screen.o csrc/prob.o csrc/engine.o csrc/parsers.o csrc/multi.o \
    -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib -o libhafgrasp_v$N.so
rm -rf $T
echo built libhafgrasp_v$N.so
