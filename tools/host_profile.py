"""cProfile of bench.py's host side (where does the Python thread spend the sampling() bracket?).
Run on the GPU box:  python tools/host_profile.py --config 4 --no-cpu-baseline --no-alt"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(45)
ps.sort_stats('tottime').print_stats(40)
print(s.getvalue()[:20000], file=sys.stderr)
