#!/bin/bash
# build a variant of the product library with extra -D flags: tools/build_variant.sh <name> <flags...> -> tools/bin/libvoxhip_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin/obj_$name
for f in kernels_lm sampler engine codec; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value "$@" -c vox_serve_amd/csrc/$f.hip -o tools/bin/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libvoxhip_$name.so tools/bin/obj_$name/*.o
echo tools/bin/libvoxhip_$name.so
