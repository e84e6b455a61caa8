"""cProfile of the host side of bench steps (run on the GPU box): where the per-launch Python time goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no_cpu_baseline", "--no_roofline"] + sys.argv[1:]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
