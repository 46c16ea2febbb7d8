#!/bin/bash
# Development build of libvoxhip with the timing knobs compiled in (-DVOX_DEV_KNOBS: VOX_ABLATE / VOX_DEV / VOX_ROWS_MIN ...,
# see csrc/engine.hip) -> tools/bin/libvoxhip_dev.so; select it with VOX_LIB=<path>.  Results are wrong when a knob is set.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/obj
for f in kernels_lm sampler engine codec; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -DVOX_DEV_KNOBS $VOX_DEV_EXTRA \
      -c vox_serve_amd/csrc/$f.hip -o tools/bin/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libvoxhip_dev.so tools/bin/obj/*.o
echo tools/bin/libvoxhip_dev.so
