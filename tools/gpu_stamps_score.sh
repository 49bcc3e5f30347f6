cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v
(cd poccala_amd/csrc && for f in pcl_api gmm_score gmm_score_mfma gmm_score_split hmm_dp gmm_accumulate gmm_accumulate_split model_derive mfcc pcl_comm; do
   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DPCL_SPLIT_STAMPS $EXTRA -c $f.hip -o ../../gpurun_out/v/$f.o 2>/dev/null; done
 hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out/v/lib_stamps.so ../../gpurun_out/v/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib)
POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/v/lib_stamps.so timeout 300 python tools/score_bench.py 1024 2048 1000 2>&1 | tail -8
