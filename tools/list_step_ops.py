"""List the torch (aten) ops one eager train step of a bench config issues, in order, with the python
source line that issued each — to find glue launches worth folding into a kernel.
usage: python tools/list_step_ops.py [cfg2|cfg3|cfg4]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    import numpy as np
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[cfg])
    dev = torch.device('cuda:0')
    agent = bench.build_agent(dev, None, bench.CFG['capacity'], 0)
    agent._use_graph = False
    bench.fill_buffer(agent, np.random.default_rng(0), 8000)
    for _ in range(5):
        agent.train()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        agent.train()
        torch.cuda.synchronize()
    # every aten op that launched a kernel itself (not through a child aten op), in launch order, with the innermost
    # python frames of this repository that issued it
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::')
           and getattr(e, 'kernels', None)
           and not any(c.name.startswith('aten::') and getattr(c, 'kernels', None) for c in (e.cpu_children or []))]
    evs.sort(key=lambda e: e.time_range.start)
    for e in evs:
        kern = e.kernels[0].name[:44]
        stack = [s_.split('/root/repo/')[-1].split('repo/')[-1] for s_ in (e.stack or []) if 'site-packages' not in s_ and 'dist-packages' not in s_
                 and '.py' in s_]
        print(f'{e.name:24s} {kern:46s} {" <- ".join(x[-60:] for x in stack[:2])}')


if __name__ == '__main__':
    main()
