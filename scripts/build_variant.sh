#!/bin/bash
# scripts/build_variant.sh NAME "EXTRA FLAGS": a tuning build of libcmdg with extra compiler flags
# into build/libcmdg_NAME.so (objects under build/obj_NAME), for scripts/ab.sh (CMDG_LIB).
set -e
name="$1"; extra="$2"
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/obj_$name
make -C $root/climatemachine.jl_amd/csrc -j8 -s OBJDIR=$root/build/obj_$name OUT=$root/build/libcmdg_$name.so EXTRA="$extra"
echo "built build/libcmdg_$name.so ($extra)"
