#!/bin/bash
# rocprofv3 evidence for bench.py's roofline block and for the accumulate pass (run on the GPU box through gpurun):
#   gpurun --timeout 2400 -- 'bash tools/gpu_profile.sh'
# one --kernel-trace --stats pass and separate --pmc passes of <= 4 counters (never combined with a trace domain), each
# under its own timeout; the summaries land in gpurun_out/prof/ and are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
ARGS="--steps 20 --warmup 2 --cpu-baseline 0 --extra 0 --sustain 0"
run() { # tag counters... (empty = trace pass), program args...
  tag=$1; shift; ctr=$1; shift
  if [ -z "$ctr" ]; then timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 "$@" > $OUT/$tag.log 2>&1 || echo "$tag failed" >> $OUT/failures.log
  else timeout 420 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$tag -- python3 "$@" > $OUT/$tag.log 2>&1 || echo "$tag ($ctr) failed" >> $OUT/failures.log; fi
}
PARTS=${PARTS:-bench acc fb dec}
export ACC_PASSES=${ACC_PASSES:-8}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has bench; then
run bench_trace "" $R/bench.py $ARGS
run bench_fetch "FETCH_SIZE" $R/bench.py $ARGS
run bench_write "WRITE_SIZE" $R/bench.py $ARGS
run bench_clk "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" $R/bench.py $ARGS
run bench_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" $R/bench.py $ARGS
run bench_sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" $R/bench.py $ARGS
fi
if has acc; then
run acc_trace "" $R/tools/acc_bench.py
run acc_fetch "FETCH_SIZE" $R/tools/acc_bench.py
run acc_write "WRITE_SIZE" $R/tools/acc_bench.py
run acc_clk "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" $R/tools/acc_bench.py
run acc_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" $R/tools/acc_bench.py
run acc_sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" $R/tools/acc_bench.py
run acc_lds "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $R/tools/acc_bench.py
run acc_tcc "TCC_HIT_sum TCC_MISS_sum" $R/tools/acc_bench.py
fi
if has fb; then
for U in 128 1024; do
run fb${U}_trace "" $R/tools/fb_bench.py $U
run fb${U}_fetch "FETCH_SIZE" $R/tools/fb_bench.py $U
run fb${U}_write "WRITE_SIZE" $R/tools/fb_bench.py $U
run fb${U}_clk "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" $R/tools/fb_bench.py $U
run fb${U}_sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $R/tools/fb_bench.py $U
run fb${U}_sq2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $R/tools/fb_bench.py $U
done
fi
DEC="$R/tools/c5_decode_bench.py 417 4096 20000 3 8192"
if has dec; then
run dec_trace "" $DEC
run dec_fetch "FETCH_SIZE" $DEC
run dec_write "WRITE_SIZE" $DEC
run dec_clk "GRBM_GUI_ACTIVE" $DEC
run dec_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" $DEC
fi
if has coarse; then
# the coarse pass on the C4 shard's model after two EM iterations (73 % off-pipe mixtures): the model is made once, outside the profiler, and left
# under /tmp of this box; every pass loads it and scores the batch four times
timeout 300 python3 $R/tools/coarse_time_probe.py make 2 > $OUT/coarse_make.log 2>&1 || echo "coarse model failed" >> $OUT/failures.log
run coarse_trace "" $R/tools/coarse_time_probe.py time
run coarse_fetch "FETCH_SIZE" $R/tools/coarse_time_probe.py time
run coarse_write "WRITE_SIZE" $R/tools/coarse_time_probe.py time
run coarse_clk "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" $R/tools/coarse_time_probe.py time
run coarse_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" $R/tools/coarse_time_probe.py time
run coarse_sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" $R/tools/coarse_time_probe.py time
run coarse_lds "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $R/tools/coarse_time_probe.py time
rm -f /tmp/em_model_*.npy
fi
RND=${RND:-r06}
python3 $R/tools/make_profile_summary.py $OUT $RND
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/failures.log 2>/dev/null
tail -22 $OUT/${RND}_bench_summary.txt; tail -22 $OUT/${RND}_accumulate_summary.txt; cat $OUT/${RND}_fb_summary.txt; tail -12 $OUT/${RND}_decode_summary.txt; has coarse && tail -30 $OUT/${RND}_coarse_summary.txt
