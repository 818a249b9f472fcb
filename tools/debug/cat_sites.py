"""Which `torch.cat` calls one eager train step of a bench config issues: shapes and the calling lines of this repository.
usage: python tools/debug/cat_sites.py cfg5"""
import sys, traceback
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
bench.CFG.clear(); bench.CFG.update(bench.CONFIGS[cfg])
dev = torch.device('cuda:0')
agent = bench.build_agent(dev, None, bench.CFG['capacity'], 0)
agent._use_graph = False
bench.fill_buffer(agent, np.random.default_rng(0), 8000)
for _ in range(4):
    agent.train()
torch.cuda.synchronize()
real_cat = torch.cat
log = []


def cat(tensors, *a, **k):
    tensors = list(tensors)
    fr = [f'{Path(f.filename).name}:{f.lineno}' for f in traceback.extract_stack()[:-1]
          if 'advanced-soft-actor-critic_amd' in f.filename or 'tests/plugins' in f.filename][-3:]
    log.append((tuple(tuple(t.shape) for t in tensors[:4]), len(tensors), a, k, ' <- '.join(reversed(fr)),
                any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor))))
    return real_cat(tensors, *a, **k)


torch.cat = cat
agent.train()
torch.cat = real_cat
for shapes, n, a, k, where, rg in log:
    print(f'{n} tensors {shapes} {a} {k} grad={rg}  {where}')
