#!/bin/bash
# The mutation builds: the library with a known, fixed bug restored must FAIL the test that guards the fix.  Build them HERE first:
#   tools/build_variant.sh lseflush "-DPCL_LSE_FLUSH_REPRO -fno-slp-vectorize" gmm_score_split.hip gmm_score_mfma.hip
#   tools/build_variant.sh race "-DPCL_DESC_RACE_REPRO" pcl_api.hip hmm_units.hip
#   tools/build_variant.sh coarsemargin "-DPCL_COARSE_MARGIN_REPRO" gmm_score_coarse.hip      (round 6: the coarse pass rules out pairs that matter)
# then:  gpurun -- 'bash tools/gpu_mutation_check.sh'.  Exit code 0 = every mutant was caught (and the shipped library passes the same tests).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rc=0
check() { # name, library, expected ("fail" / "pass"), pytest args...
  name=$1; lib=$2; want=$3; shift 3
  POCCALA_HIP_LIB=$lib timeout -k 10 400 python3 -m pytest "$@" -q > gpurun_out/mutation_$name.txt 2>&1; got=$?
  if [ "$want" = fail ] && [ $got -ne 0 ] && grep -q "failed" gpurun_out/mutation_$name.txt; then echo "$name: CAUGHT ($(tail -1 gpurun_out/mutation_$name.txt))";
  elif [ "$want" = pass ] && [ $got -eq 0 ]; then echo "$name: passes ($(tail -1 gpurun_out/mutation_$name.txt))";
  else echo "$name: UNEXPECTED rc=$got ($(tail -1 gpurun_out/mutation_$name.txt))"; rc=1; fi
}
L=$GRAFT_REPO_ROOT/poccala_amd/libpoccala_hip.so
check shipped_lse $L pass tests/test_gpu_parity.py -k "far_above_the_first_tile"
check mutant_lse $GRAFT_REPO_ROOT/build_ab/lib_lseflush.so fail tests/test_gpu_parity.py -k "far_above_the_first_tile"
check shipped_race $L pass tests/test_gpu_sweep.py
check mutant_race $GRAFT_REPO_ROOT/build_ab/lib_race.so fail tests/test_gpu_sweep.py
check shipped_coarse $L pass tests/test_gpu_coarse.py -k "six_decades or e_step"
check mutant_coarse $GRAFT_REPO_ROOT/build_ab/lib_coarsemargin.so fail tests/test_gpu_coarse.py -k "six_decades or e_step"
exit $rc
