"""cProfile of the per-utterance drop-in worker (bench.zero_change_route): where the host time of the zero-change route goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
bench.zero_change_route(1)
pr = cProfile.Profile(); pr.enable()
r = bench.zero_change_route(2)
pr.disable()
print(r['frames_per_s'], r['ms_per_utterance'])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
