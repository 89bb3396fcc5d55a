#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-DFLAG ...]: build libcine_hip_NAME.so with one translation unit recompiled under extra flags
# (A/B kernels on the GPU box through CINE_HIP_LIB=...).  Output: deep-cine-cardiac-mri_amd/csrc/build/variants/.
set -e
cd "$(dirname "$0")/../deep-cine-cardiac-mri_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on "$@" -c $src -o build/variants/$name.$src.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/variants/$name.$src.o -o build/variants/libcine_hip_$name.so
echo build/variants/libcine_hip_$name.so
