"""Which Python line launches which kernel of a train step?  One EAGER step of a bench configuration under torch.profiler
(with stacks): the ATen kernels in launch order, each with the innermost frames of this repository that issued it.  The
framework's own launches (ctypes into libasac_hip.so) carry no ATen op: they show up between them by name only.

    python tools/trace_step.py cfg5 [out.txt]
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    name = sys.argv[1]
    out_path = sys.argv[2] if len(sys.argv) > 2 else f'gpurun_out/trace_{name}.txt'
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[name])
    import os
    os.environ['ASAC_BENCH_HIP_CONFIG'] = '{"use_graph": false}'
    agent = bench.build_agent('cuda:0', None, bench.CFG['capacity'], 0)
    bench.fill_buffer(agent, np.random.default_rng(0), 8192)
    for _ in range(3):
        agent.train()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        agent.train()
        torch.cuda.synchronize()
    def own(stack):
        frames = [f for f in (stack or []) if 'advanced-soft-actor-critic_amd' in f or 'tests/plugins' in f]
        return ' <- '.join(f.split('advanced-soft-actor-critic_amd/')[-1].split('/root/repo/')[-1] for f in frames[:3])

    events = list(prof.events())
    forward_of = {}       # autograd sequence number -> the frames of the forward op that recorded the node
    for ev in events:
        if ev.sequence_nr is not None and ev.sequence_nr >= 0 and own(ev.stack) and ev.sequence_nr not in forward_of:
            forward_of[ev.sequence_nr] = own(ev.stack)
    rows = []
    for ev in events:
        if not ev.kernels:
            continue
        where = own(ev.stack)
        if not where and ev.sequence_nr is not None and ev.sequence_nr in forward_of:
            where = '(backward of) ' + forward_of[ev.sequence_nr]
        node, up = '', ev.cpu_parent
        while up is not None:       # the autograd node (backward) or the module-level op this ATen call serves
            if up.name.startswith('autograd::engine::evaluate_function: '):
                node = up.name.split(': ', 1)[1]
                break
            up = up.cpu_parent
        for k in ev.kernels:
            rows.append((ev.time_range.start, k.name[:60], k.duration, ev.name + (f'  [{node}]' if node else ''), where))
    rows.sort()
    Path(out_path).parent.mkdir(parents=True, exist_ok=True)
    with open(out_path, 'w') as f:
        f.write(f'# {name}: {len(rows)} ATen-issued kernels of one eager step (launch order)\n')
        for _, kname, dur, op, where in rows:
            f.write(f'{dur:8.1f} us  {kname:44.44s}  {op:70s}  {where}\n')
    print(f'wrote {out_path}: {len(rows)} kernels')


if __name__ == '__main__':
    main()
