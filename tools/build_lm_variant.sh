#!/bin/bash
# variant of the product library that differs in ONE source only (default kernels_lm.hip), the other objects taken from the last
# product build: tools/build_lm_variant.sh <name> [-DFLAG ...] [SRC=codec]  -> tools/bin/libvoxhip_<name>.so   (run several in parallel)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=kernels_lm; flags=()
for a in "$@"; do case $a in SRC=*) src=${a#SRC=};; *) flags+=("$a");; esac; done
mkdir -p tools/bin/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "${flags[@]}" -c vox_serve_amd/csrc/$src.hip -o tools/bin/obj_$name/$src.o
objs=()
for f in kernels_lm sampler engine codec; do
  if [ $f = $src ]; then objs+=(tools/bin/obj_$name/$f.o); else objs+=(vox_serve_amd/build/$f.hip.o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libvoxhip_$name.so "${objs[@]}"
echo tools/bin/libvoxhip_$name.so
