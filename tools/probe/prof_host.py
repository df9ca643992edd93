import cProfile, pstats, sys, os, io
sys.argv = ['bench.py', '--no-extras', '--no-cpu-baseline', '--steps', '30']
sys.path.insert(0, os.getcwd())
import bench
pr = cProfile.Profile()
orig = bench.timed_steps
calls = [0]
def wrapped(*a, **k):
    calls[0] += 1
    if calls[0] == 2:       # the headline region (the first call is the setup)
        pr.enable()
        r = orig(*a, **k)
        pr.disable()
        return r
    return orig(*a, **k)
bench.timed_steps = wrapped
try:
    bench.main()
except SystemExit:
    pass
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:5000], file=sys.stderr)
