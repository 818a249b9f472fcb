"""Is the gap between two graph replays smaller on a non-default stream?  (cfg2, plain train() calls)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

bench.CFG.clear()
bench.CFG.update(bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
for mode in ('default', 'side', 'default', 'side'):
    agent = bench.build_agent(dev, None, bench.CFG['capacity'], seed=0)
    bench.fill_buffer(agent, np.random.default_rng(0), 2 ** 16)
    s = torch.cuda.Stream() if mode == 'side' else torch.cuda.current_stream()
    with torch.cuda.stream(s):
        for _ in range(300):
            agent.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4000):
            agent.train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(mode, round(4000 / dt, 1), 'steps/s', 'graph' if agent._graph is not None else 'eager', flush=True)
    agent.close()
