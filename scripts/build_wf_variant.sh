#!/bin/bash
# scripts/build_wf_variant.sh NAME "<extra flags for wavefront.hip>"  -> ray-tracer_amd/lib_variants/NAME/librtx_hip.so
# (tuning experiments: run with RTX_HIP_LIB=ray-tracer_amd/lib_variants/NAME/librtx_hip.so)
set -e
cd "$(dirname "$0")/../ray-tracer_amd/csrc"
NAME=$1; FLAGS=$2
OUT=../lib_variants/$NAME; mkdir -p $OUT ../build
make -s -j8 >/dev/null 2>&1 || make -s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops -fno-hip-fp32-correctly-rounded-divide-sqrt -funsafe-math-optimizations -fgpu-flush-denormals-to-zero $FLAGS -c wavefront.hip -o ../build/wf_$NAME.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librtx_hip.so ../build/kernels_f64.o ../build/kernels_f32.o ../build/wf_$NAME.o ../build/bvh_build.o ../build/pt_host.o ../build/rtx_api.o -lpthread
echo built $OUT
