#!/bin/bash
# Builds whole-library variants for scripts/lab/ab_libs.sh: scripts/lab/_ab/<name>.so = the library as built (vectordb_amd/lib/obj/*.o)
# with mfma_filter.hip recompiled under extra -D switches.   usage: build_variants.sh name1="-DEPS_V7_VI=6" name2="-D..." ...
# (run `python -m vectordb_amd.build` first; hipcc cross-compiles, no GPU needed)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/scripts/lab/_ab
OBJ=$R/vectordb_amd/lib/obj
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  [ "$defs" == "$spec" ] && defs=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I$R/include $defs \
      -c $R/vectordb_amd/csrc/mfma_filter.hip -o $R/scripts/lab/_ab/mfma_filter_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/lab/_ab/$name.so $OBJ/index.o $OBJ/shard_group.o $OBJ/exchange.o $OBJ/flat_kernels.o $OBJ/traverse.o \
      $R/scripts/lab/_ab/mfma_filter_$name.o $OBJ/graph_build.o -ldl
  echo "built scripts/lab/_ab/$name.so ($defs)"
done
