#!/bin/bash
# tools/build_stamps.sh [-DFLAG ...]: build tools/tmp/conv_stamps.bin (phase stamps of one conv3x3 / transpose-conv layer)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=on -I deep-cine-cardiac-mri_amd/csrc "$@" -c tools/conv_stamps.hip -o tools/tmp/conv_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/tmp/conv_stamps.o deep-cine-cardiac-mri_amd/csrc/build/api.cpp.o deep-cine-cardiac-mri_amd/csrc/build/conv_plane.hip.o deep-cine-cardiac-mri_amd/csrc/build/conv_coarse.hip.o deep-cine-cardiac-mri_amd/csrc/build/grad_kernels.hip.o deep-cine-cardiac-mri_amd/csrc/build/inbwd_fast.hip.o -o tools/tmp/conv_stamps.bin   # (the general kernel's dispatcher refers to the lean / coarse launchers)
