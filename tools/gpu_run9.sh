cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --cpu-baseline 0 > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3g frames/s  ms/step %.2f  score %.2f ms (%.1f TF, frac %.3f)  fb %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['fb_kernel_avg_ms']))
print(d['extra'])"
