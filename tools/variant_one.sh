#!/bin/bash
# A library variant that differs from the in-tree build in ONE translation unit (seconds instead of a minute):
#   tools/variant_one.sh <name> <file.hip> <extra compiler flags...>   ->  tools/bin/libzafx_<name>.so   (run `make` in csrc first)
set -e
name=$1; file=$2; shift; shift
cd "$(dirname "$0")/../zaf-python_amd/csrc"
mkdir -p ../../tools/bin/obj_${name}
obj=../../tools/bin/obj_${name}/${file%.*}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast "$@" -c $file -o $obj
others=$(ls build/*.o | grep -v "/${file%.*}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libzafx_${name}.so $obj $others -ldl
