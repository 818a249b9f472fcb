import sys, json
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import bench
bench.CFG.clear(); bench.CFG.update(bench.CONFIGS['cfg5'])
from asac_amd import native
agent = bench.build_agent('cuda:0', None, 65536, 0)
agent._use_graph = False
rng = np.random.default_rng(0)
bench.fill_buffer(agent, rng, 4000)
for _ in range(3): agent.train()
with native.LaunchProfiler(repeat=1) as prof:
    agent.train()
s = prof.summary()
print({k: v['calls'] for k, v in s.items() if 'conv' in k})
print('single', agent._rpm_single_backward, 'one', agent._rpm_conv_one_launch, 'disjoint', agent._rpm_models_disjoint())
