"""Where do the ~9 us between two replays of the captured cfg2 step go?  Host time of a `train()` call and of the bare
`asac_graph_launch` against the device time of the step: if the host loop finishes long before the device, the gap is the
GPU's (command processor / barrier packets between graphs), otherwise the launch path is the bottleneck."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from asac_amd import native  # noqa: E402

bench.CFG.clear()
bench.CFG.update(bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
agent = bench.build_agent(dev, None, bench.CFG['capacity'], seed=0)
bench.fill_buffer(agent, np.random.default_rng(0), 2 ** 16)
for _ in range(300):
    agent.train()
torch.cuda.synchronize()
N = 4000
for name, fn in (('train()', agent.train),
                 ('bare asac_graph_launch', (lambda: native.graph_launch(agent._graph_exec)) if getattr(agent, '_graph_exec', None) else None)):
    if fn is None:
        print(name, 'n/a (no raw exec handle)')
        continue
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
        print(f'{name:26s} host loop {1e6 * t_host / N:7.2f} us/call, until device idle {1e6 * t_dev / N:7.2f} us/call '
              f'({N / t_dev:8.1f} steps/s)', flush=True)
# host cost of one call with an idle device (nothing queued): launch + immediate sync
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((t1 - t0, time.perf_counter() - t0))
ts = np.array(ts) * 1e6
print(f'idle device: train() returns after {np.median(ts[:, 0]):.2f} us, device done after {np.median(ts[:, 1]):.2f} us')
agent.close()
