#!/bin/bash
# Build tools/bin/libzafx_prof.so: the library with the per-phase cycle counters compiled in.
set -e
cd "$(dirname "$0")/../zaf-python_amd/csrc"
mkdir -p ../../tools/bin/prof
for f in zafx_capi.cpp zafx_stft.hip zafx_mdct.hip zafx_mel.hip zafx_cqt.hip zafx_pcm.hip zafx_linear.hip zafx_dct.hip zafx_f64.hip zafx_bs32.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DZAFX_PROF -x hip -c $f -o ../../tools/bin/prof/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libzafx_prof.so ../../tools/bin/prof/*.o -ldl
