#!/bin/bash
# scripts/build_f32_variant.sh NAME "<extra flags for kernels_f32.hip>"  -> ray-tracer_amd/lib_variants/NAME/librtx_hip.so
set -e
cd "$(dirname "$0")/../ray-tracer_amd/csrc"
NAME=$1; FLAGS=$2
OUT=../lib_variants/$NAME; mkdir -p $OUT ../build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-hip-fp32-correctly-rounded-divide-sqrt -funsafe-math-optimizations $FLAGS -c kernels_f32.hip -o ../build/f32_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librtx_hip.so ../build/kernels_f64.o ../build/f32_$NAME.o ../build/wavefront.o ../build/pt_host.o ../build/rtx_api.o -lpthread
echo built $OUT
