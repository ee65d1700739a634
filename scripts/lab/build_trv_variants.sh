#!/bin/bash
# Whole-library variants for traversal A/Bs: scripts/lab/_ab/<name>.so = the library as built (vectordb_amd/lib/obj/*.o) with traverse.hip
# recompiled under extra -D switches.   usage: build_trv_variants.sh name1="-DEPS_TRV_FUSED=0" name2="-D..." ...
# (run `python -m vectordb_amd.build` first; hipcc cross-compiles, no GPU needed)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/scripts/lab/_ab
OBJ=$R/vectordb_amd/lib/obj
for spec in "$@"; do
  (
  name=${spec%%=*}; defs=${spec#*=}
  [ "$defs" == "$spec" ] && defs=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -I$R/include $defs \
      -c $R/vectordb_amd/csrc/traverse.hip -o $R/scripts/lab/_ab/traverse_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/lab/_ab/$name.so $OBJ/index.o $OBJ/shard_group.o $OBJ/flat_kernels.o \
      $R/scripts/lab/_ab/traverse_$name.o $OBJ/mfma_filter.o $OBJ/graph_build.o
  echo "built scripts/lab/_ab/$name.so ($defs)"
  ) &
done
wait
