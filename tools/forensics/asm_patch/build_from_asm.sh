#!/bin/bash
# usage: build_from_asm.sh dev.s out_tag   -> /root/repo/video-frame-interpolation_amd/emavfi/lib/libemavfi_<tag>.so
set -e
S=$1; TAG=$2
LLVM=/opt/rocm/lib/llvm/bin
CS=/root/repo/video-frame-interpolation_amd/csrc
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $S -o dev_$TAG.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev_$TAG.out dev_$TAG.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev_$TAG.out -output=dev_$TAG.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DEMAVFI_WARP_DIVERGENT=1 -I$CS -I/root/repo/include --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang dev_$TAG.hipfb -c $CS/misc_kernels.hip -o misc_$TAG.o
B=/root/repo/build/csrc_div
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/video-frame-interpolation_amd/emavfi/lib/libemavfi_$TAG.so $B/emavfi_api.o misc_$TAG.o $B/conv3x3_f32.o $B/conv3x3_bf16.o $B/conv3x3_f16.o $B/deform_f32.o $B/deform_bf16.o $B/deform_f16.o
echo built libemavfi_$TAG.so
