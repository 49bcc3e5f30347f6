# A/B builds of the split accumulate kernel on the GPU box: each argument is a set of -D flags
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v
build() { # name flags
  (cd poccala_amd/csrc && for f in pcl_api gmm_score gmm_score_mfma gmm_score_split hmm_dp gmm_accumulate gmm_accumulate_split model_derive mfcc pcl_comm; do SLP=""; [ $f = gmm_score_split ] && SLP=-fno-slp-vectorize;
     if [ $f = gmm_accumulate_split ] || [ ! -f ../../gpurun_out/v/$f.o ]; then hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $SLP $2 -c $f.hip -o ../../gpurun_out/v/$f.o 2>/dev/null; fi; done
   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out/v/lib_$1.so ../../gpurun_out/v/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib)
}
run() { POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/v/lib_$1.so timeout 300 python tools/acc_bench.py 2>&1 | tail -1 >> gpurun_out/acc_ab.log; }
rm -f gpurun_out/acc_ab.log gpurun_out/v/*
for v in "$@"; do name=$(echo "$v" | tr -d ' =-' | tr -c 'A-Za-z0-9\n' '_'); build "x$name" "$v"; run "x$name"; done
cat gpurun_out/acc_ab.log
