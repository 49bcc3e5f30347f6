cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -W ignore -x 2>&1 | tail -4
timeout 600 python tools/c5_decode_bench.py 1668 4096 20000 4 8192 2>&1 | grep -v "^tree\|^utterance\|^decode ~" | cut -c1-330
timeout 600 python bench.py --cpu-baseline 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); e=d['extra']
print(d['value'], d['ms_per_step'], e['estep_ms'], e['accumulate_ms'], e['zero_change_route']['frames_per_s'], e['setup_s'])"
