"""Host-side timing of plain train() calls in chunks (is a process's timed region uniformly slow, or only its start?):
python tools/debug/step_chunks.py cfg4 [chunks] [steps per chunk]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
chunks, per = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10, 50)
bench.CFG.clear()
bench.CFG.update(bench.CONFIGS[cfg])
device = torch.device('cuda', 0)
torch.cuda.set_device(device)
from asac_amd import native  # noqa: E402
native.load()
agent = bench.build_agent(device, None, bench.CFG['capacity'], seed=0)
bench.fill_buffer(agent, np.random.default_rng(1234), bench.CFG['fill'])
for _ in range(agent._graph_warmup + 43):
    agent.train()
torch.cuda.synchronize()
out = []
for c in range(chunks):
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(per):
        h0 = time.perf_counter()
        agent.train()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    out.append((1e6 * (time.perf_counter() - t0) / per, 1e6 * host / per))
print(cfg, 'graph', agent._graph is not None, ' us/step (wall, host inside train()):', ' '.join(f'{a:.0f}/{b:.0f}' for a, b in out))
