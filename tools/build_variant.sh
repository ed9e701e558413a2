#!/bin/bash
# Build a variant of the library under tools/bin/ for A/B runs through ZAFX_LIBRARY (or tools/placement.py):
#   tools/build_variant.sh <name> <extra compiler flags...>      ->  tools/bin/libzafx_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../zaf-python_amd/csrc"
mkdir -p ../../tools/bin
make -j8 OUT=../../tools/bin/libzafx_${name}.so OBJDIR=../../tools/bin/obj_${name} EXTRA="$*" all
