#!/bin/bash
# Experiment builds of libdivans_hip.so with the decode2 switches flipped (gpurun_exp/*.so travel to the GPU box; *.so is git-ignored).
# usage: scripts/build_variants.sh   then   DIVANS_HIP_LIBRARY=gpurun_exp/libdivans_noasync.so python scripts/decode2_sweep.py ...
set -e
cd "$(dirname "$0")/.."
python divans_amd/build.py > /dev/null
mkdir -p gpurun_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJS=$(ls divans_amd/build/*.o | grep -v lit_decode2 | grep -v check | grep -v lit_decode_t)
build () {  # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -x hip -c divans_amd/csrc/lit_decode2.hip -o gpurun_exp/lit_decode2_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_$1.so $OBJS gpurun_exp/lit_decode2_$1.o
  rm gpurun_exp/lit_decode2_$1.o
}
if [ "$1" = "noperm" ]; then
  build noperm "-DDIVANS_D2_PERM=0" &
elif [ "$1" = "hs_rows" ]; then    # the high-nibble stride table as [ctx][prev] (round 3) instead of [class][prev] (capi.cpp derive_geometry)
  OBJS2=$(ls divans_amd/build/*.o | grep -v capi)
  /opt/rocm/bin/hipcc $FLAGS -DDIVANS_HS_BY_CLASS=0 -x hip -c divans_amd/csrc/capi.cpp -o gpurun_exp/capi_hsctx.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_hsctx.so $OBJS2 gpurun_exp/capi_hsctx.o
  rm gpurun_exp/capi_hsctx.o
elif [ "$1" = "rans2" ]; then      # workgroup size of the chunk-parallel rANS pass (lit_kernels.hip): one wave per workgroup as in round 3
  OBJS2=$(ls divans_amd/build/*.o | grep -v lit_kernels)
  /opt/rocm/bin/hipcc $FLAGS -DDIVANS_RANS2_THREADS=64 -x hip -c divans_amd/csrc/lit_kernels.hip -o gpurun_exp/lit_kernels_rans64.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_rans64.so $OBJS2 gpurun_exp/lit_kernels_rans64.o
  rm gpurun_exp/lit_kernels_rans64.o
elif [ "$1" = "dm_auto" ]; then    # the batch ABI with and without the per-batch choice of direct-mapped caches (capi.cpp)
  OBJS2=$(ls divans_amd/build/*.o | grep -v capi)
  /opt/rocm/bin/hipcc $FLAGS -DDIVANS_DM_AUTO_DEFAULT=0 -x hip -c divans_amd/csrc/capi.cpp -o gpurun_exp/capi_nodmauto.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_nodmauto.so $OBJS2 gpurun_exp/capi_nodmauto.o
  rm gpurun_exp/capi_nodmauto.o
elif [ "$1" = "switches" ]; then   # capi.cpp with the measurement switches that release builds compile out (DIVANS_TABLES_ALLOC, DIVANS_SLAB_ROWS_MOD, DIVANS_DEBUG_ALLOC):
  OBJS2=$(ls divans_amd/build/*.o | grep -v capi | grep -v check)   # what scripts/placement_probe.py, placement_map.py, placement_counters*.sh need
  /opt/rocm/bin/hipcc $FLAGS -DDIVANS_EXPERIMENT_SWITCHES=1 -x hip -c divans_amd/csrc/capi.cpp -o gpurun_exp/capi_switches.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_switches.so $OBJS2 gpurun_exp/capi_switches.o
  rm gpurun_exp/capi_switches.o
elif [ "$1" = "rans_old" ]; then   # round 1-4's 64-bit division in the rANS pass (lit_kernels.hip), for scripts/r05c_rans_ab.sh
  OBJS2=$(ls divans_amd/build/*.o | grep -v lit_kernels | grep -v check)
  /opt/rocm/bin/hipcc $FLAGS -DDIVANS_RANS_DIVMOD=0 -x hip -c divans_amd/csrc/lit_kernels.hip -o gpurun_exp/lit_kernels_ransold.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libdivans_ransold.so $OBJS2 gpurun_exp/lit_kernels_ransold.o
  rm gpurun_exp/lit_kernels_ransold.o
elif [ "$1" = "pad_valu" ]; then   # +N four-cycle VALU instructions per byte: how much of the time is VALU issue?
  build pad10 "-DDIVANS_D2_PAD_VALU=10" &
  build pad26 "-DDIVANS_D2_PAD_VALU=26" &
  build pad52 "-DDIVANS_D2_PAD_VALU=52" &
elif [ "$1" = "cache_policy" ]; then
  build ldnt "-DDIVANS_D2_LOAD_AUX=2" &
  build stnt "-DDIVANS_D2_STORE_AUX=2" &
  build ldstnt "-DDIVANS_D2_LOAD_AUX=2 -DDIVANS_D2_STORE_AUX=2" &
  build stsc1 "-DDIVANS_D2_STORE_AUX=16" &
  build ldsc0 "-DDIVANS_D2_LOAD_AUX=1" &
else
  build noasync "-DDIVANS_D2_ASYNC=0" &
  build now7 "-DDIVANS_D2_W7=0" &
  build noasync_now7 "-DDIVANS_D2_ASYNC=0 -DDIVANS_D2_W7=0" &
fi
wait
ls -la gpurun_exp
