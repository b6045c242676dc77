#!/bin/bash
# A/B builds of the screening kernel: tools/build_variant.sh N [extra -D flags] -> haf_grasping_amd/libhafgrasp_v N.so
# (screen.hip compiled with -DSCREEN_VARIANT=N, everything else from the regular build).  Time them on ONE GPU box with
#   HAF_LIB=haf_grasping_amd/libhafgrasp_vN.so python tools/time_svm_stage.py
set -e
cd "$(dirname "$0")/../haf_grasping_amd"
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-inline-asm -fno-slp-vectorize \
    -DSCREEN_VARIANT=$N "$@" -c csrc/screen.hip -o /tmp/screen_v$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC csrc/kernels.o /tmp/screen_v$N.o csrc/prob.o csrc/engine.o csrc/parsers.o csrc/multi.o \
    -L/opt/rocm/lib -lrccl -lpthread -Wl,-rpath,/opt/rocm/lib -o libhafgrasp_v$N.so
echo built libhafgrasp_v$N.so
