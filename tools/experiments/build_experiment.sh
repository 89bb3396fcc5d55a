#!/bin/bash
# tools/experiments/build_experiment.sh NAME [-DFLAG ...]: apply r03_rejected_paths.patch to a scratch copy of csrc/, add unet_bottom.hip,
# and build libcine_hip_NAME.so from it (csrc/build/variants/).  The product sources are not touched.
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
name=$1; shift
work=$(mktemp -d)
mkdir -p "$work/deep-cine-cardiac-mri_amd" "$work/include"
cp -r "$root/deep-cine-cardiac-mri_amd/csrc" "$work/deep-cine-cardiac-mri_amd/csrc"
cp "$root/include/cine_hip.h" "$work/include/"
rm -rf "$work/deep-cine-cardiac-mri_amd/csrc/build"
(cd "$work" && git init -q . && git apply --whitespace=nowarn "$here/r03_rejected_paths.patch")
cp "$here/unet_bottom.hip" "$work/deep-cine-cardiac-mri_amd/csrc/"
cd "$work/deep-cine-cardiac-mri_amd/csrc"
objs=""
for f in *.hip api.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on "$@" -x hip -c $f -o $f.o &
  objs="$objs $f.o"
done
wait
mkdir -p "$root/deep-cine-cardiac-mri_amd/csrc/build/variants"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "$root/deep-cine-cardiac-mri_amd/csrc/build/variants/libcine_hip_$name.so"
echo "$root/deep-cine-cardiac-mri_amd/csrc/build/variants/libcine_hip_$name.so"
rm -rf "$work"
