#!/bin/bash
# usage: tools/probes/x6_build.sh <tag> "<extra hipcc flags>"   -> build/x6/libvit_<tag>.so (kernel experiment build)
cd "$(dirname "$0")/../../styl3r_amd/csrc"
mkdir -p ../../build/x6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden $2 vit_rope.hip vit_attention.hip vit_attention_tail.hip vit_attention_bwd.hip vit_gemm.hip vit_gemm_x6.hip vit_resample.hip vit_api.hip -o ../../build/x6/libvit_$1.so && echo built $1
