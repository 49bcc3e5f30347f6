#!/bin/bash
# Build an A/B variant of libpoccala_hip.so HERE (hipcc cross-compiles; built .so files travel to the GPU box):
#   tools/build_variant.sh NAME "FLAGS" file1.hip [file2.hip ...]   -> build_ab/lib_NAME.so
# The named sources are recompiled with FLAGS, every other object comes from the regular build.  Select on the GPU box
# with POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_NAME.so.
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C poccala_amd/csrc
name=$1; flags=$2; shift 2
mkdir -p build_ab/$name
objs=""
for o in poccala_amd/csrc/*.o; do
  b=$(basename $o .o); keep=1
  for f in "$@"; do [ "$(basename $f .hip)" = "$b" ] && keep=0; done
  [ $keep = 1 ] && objs="$objs $o"
done
for f in "$@"; do
  b=$(basename $f .hip); slp=""; [ $b = gmm_score_split ] && slp=-fno-slp-vectorize
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $slp $flags -c poccala_amd/csrc/$b.hip -o build_ab/$name/$b.o
  objs="$objs build_ab/$name/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$name.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo build_ab/lib_$name.so
