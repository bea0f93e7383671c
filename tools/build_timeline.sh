#!/bin/bash
# Experiment build: libicap_tl.so = libicap.so with gemm_bf16.hip, gemm_f32.hip and persistent.hip
# compiled under -DICAP_TIMELINE.
set -e
cd "$(dirname "$0")/../image_caption_amd/csrc"
mkdir -p build_tl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DICAP_TIMELINE=${ICAP_TIMELINE:-1} -c gemm_bf16.hip -o build_tl/gemm_bf16.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DICAP_TIMELINE=${ICAP_TIMELINE:-1} -c gemm_f32.hip -o build_tl/gemm_f32.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DICAP_TIMELINE=${ICAP_TIMELINE:-1} -c persistent.hip -o build_tl/persistent.o
objs=$(ls build/*.o | grep -v gemm_bf16.o | grep -v gemm_f32.o | grep -v persistent.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build_tl/gemm_bf16.o build_tl/gemm_f32.o build_tl/persistent.o -o ../libicap_tl.so
