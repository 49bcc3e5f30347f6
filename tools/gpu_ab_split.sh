# A/B builds of the split-bf16 scoring kernel on the GPU box: each argument is a set of -D flags
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v
build() { # name flags
  (cd poccala_amd/csrc && for f in pcl_api gmm_score gmm_score_mfma gmm_score_split hmm_dp gmm_accumulate gmm_accumulate_split model_derive mfcc pcl_comm; do
     if [ $f = gmm_score_split ] || [ $f = gmm_score ] || [ $f = pcl_api ] || [ ! -f ../../gpurun_out/v/$f.o ]; then hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $( [ $f = gmm_score_split ] && echo -fno-slp-vectorize ) $2 -c $f.hip -o ../../gpurun_out/v/$f.o 2>/dev/null; fi; done
   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out/v/lib_$1.so ../../gpurun_out/v/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib)
}
run() { POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/v/lib_$1.so PCL_SCORE_VARIANT=${VARIANT:-5} CHECK=1 timeout 300 python tools/score_bench.py 1024 2048 1000 2>&1 | tail -2 >> gpurun_out/split_ab.log; }
rm -f gpurun_out/split_ab.log gpurun_out/v/*
for v in "$@"; do name=$(echo "$v" | tr -d ' =-' | tr -c 'A-Za-z0-9\n' '_'); build "x$name" "$v"; run "x$name"; done
cat gpurun_out/split_ab.log
