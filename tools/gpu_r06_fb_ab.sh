#!/bin/bash
cd $GRAFT_REPO_ROOT
for lib in "$GRAFT_REPO_ROOT/poccala_amd/libpoccala_hip.so" "$GRAFT_REPO_ROOT/build_ab/lib_fbsites.so"; do
  echo "== $lib"
  for args in "1024 20" "128 20" "1024 21" "128 40" "1024 84"; do
    for rep in 1 2; do POCCALA_HIP_LIB=$lib timeout -k 10 120 python3 tools/fb_bench.py $args 2>&1 | grep "fix_pi=False" | sed 's/, logP.*//'; done
  done
done
