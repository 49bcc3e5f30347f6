cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v
build() { # name flags
  (cd poccala_amd/csrc && for f in pcl_api gmm_score hmm_dp gmm_accumulate pcl_comm; do
     if [ $f = gmm_score ] || [ ! -f ../../gpurun_out/v/$f.o ]; then hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $2 -c $f.hip -o ../../gpurun_out/v/$f.o 2>/dev/null; fi; done
   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_out/v/lib_$1.so ../../gpurun_out/v/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib)
}
run() { POCCALA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_out/v/lib_$1.so PCL_SCORE_VARIANT=${2:-1} CHECK=1 timeout 120 python tools/score_bench.py >> gpurun_out/score_ab.log 2>&1; }
rm -f gpurun_out/score_ab.log
build base ""; run base 1; run base 2
build r3 "-DPCL_R32=3"; run r3
build r5 "-DPCL_R32=5"; run r5
build g2 "-DPCL_GROUP=2"; run g2
build ch128 "-DPCL_CH32=128"; run ch128
build r3ch128 "-DPCL_R32=3 -DPCL_CH32=128"; run r3ch128
cat gpurun_out/score_ab.log
hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu 2>/dev/null && timeout 120 ./gpurun_out/ubench_valu > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
