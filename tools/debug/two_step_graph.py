"""Experiment: what does the boundary between two graph launches cost?  Capture ONE and TWO train steps per graph and
time the replays (events around 4000 steps)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from asac_amd import native

bench.CFG.clear(); bench.CFG.update(bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
ag = bench.build_agent(dev, None, bench.CFG['capacity'], seed=0)
bench.fill_buffer(ag, np.random.default_rng(0), bench.CFG['fill'])
for _ in range(50):
    ag.train()
torch.cuda.synchronize()
assert ag._graph is not None
for k in (1, 2, 4):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
        for _ in range(k):
            ag._device_step()
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    ex = int(g.raw_cuda_graph_exec())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 4000 // k
    e0.record()
    for _ in range(n):
        native.graph_launch(ex)
    e1.record()
    torch.cuda.synchronize()
    print(f'{k} step(s) per graph: {e0.elapsed_time(e1) * 1e3 / (n * k):.2f} us per step')
