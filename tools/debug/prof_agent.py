import cProfile, pstats, sys, io
sys.path.insert(0, '.')
sys.argv = ['agent_bench']
import tools.agent_bench as ab
import torch
pr = cProfile.Profile()
ab.run('vec', 8, 50)      # warm
pr.enable()
out = ab.run('vec', 8, 300)
pr.disable()
print(out)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
