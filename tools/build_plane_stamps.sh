#!/bin/bash
# tools/build_plane_stamps.sh [-DFLAG ...]: build tools/tmp/plane_stamps.bin (phase stamps of one lean plane-conv layer)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/tmp
C=deep-cine-cardiac-mri_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value -ffp-contract=on -I $C "$@" -c tools/plane_stamps.hip -o tools/tmp/plane_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/tmp/plane_stamps.o $C/build/conv_kernels.hip.o $C/build/conv_coarse.hip.o $C/build/api.cpp.o -o tools/tmp/plane_stamps.bin
