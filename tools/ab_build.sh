#!/bin/bash
# Builds the library from git HEAD's kernel sources next to the working tree's (build/ab/liborbx_head.so) so that tools/ab_env.sh can compare
# two CODE states inside one gpurun call (ORBX_LIBRARY=build/ab/liborbx_head.so; devices of the pool differ by several per cent, so numbers
# from two calls do not compare).   usage (here, before gpurun): tools/ab_build.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
rm -rf $R/build/ab/src && mkdir -p $R/build/ab/src/extractorb_amd $R/build/ab/src/build
git -C $R archive HEAD extractorb_amd/csrc include | tar -x -C $R/build/ab/src
make -C $R/build/ab/src/extractorb_amd/csrc OUT=$R/build/ab/liborbx_head.so OBJDIR=$R/build/ab/obj 2>&1 | grep -E " error|liborbx_head" || true
ls -la $R/build/ab/liborbx_head.so
