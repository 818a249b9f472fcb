"""Phase timeline of the step's last launch (`k_td_update`: temperature step + TD errors' return + priority update, one
workgroup) inside the real cfg2 step.  Builds a copy of the library with -DASAC_TD_STAMPS (100 MHz wall-clock stamps of
thread 0), runs eager train steps and prints the mean time between stamps.   (on a GPU box)  python tools/td_phases.py"""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CSRC = ROOT / 'advanced-soft-actor-critic_amd' / 'csrc'
LIB = ROOT / 'advanced-soft-actor-critic_amd' / 'lib'


def build():
    out = LIB / 'libasac_hip_tdstamps.so'
    obj = LIB / 'obj' / 'returns_stamps.o'
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-function',
             f'-I{ROOT / "include"}', f'-I{CSRC}']
    flags += [f'-D{d}' for d in os.environ.get('TD_PHASES_DEFINES', '').split()]
    subprocess.check_call(['/opt/rocm/bin/hipcc', *flags, '-DASAC_TD_STAMPS', '-c', str(CSRC / 'returns.hip'), '-o', str(obj)])
    objs = [str(p) for p in (LIB / 'obj').glob('*.o') if p.name not in ('returns.o', 'returns_stamps.o')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, str(obj), '-o', str(out)])
    return out


def main():
    lib_path = build()
    os.environ['ASAC_HIP_LIB'] = str(lib_path)
    import numpy as np
    import torch
    import bench
    from asac_amd import native
    lib = native.load()
    agent = bench.build_agent(torch.device('cuda:0'), None, bench.CFG['capacity'], seed=0)
    agent._use_graph = False
    bench.fill_buffer(agent, np.random.default_rng(0), 20000)
    for _ in range(20):
        agent.train()
    raw = ctypes.CDLL(str(lib_path))
    acc, reps = np.zeros(8), 200
    for _ in range(reps):
        agent.train()
        torch.cuda.synchronize()
        st = (ctypes.c_ulonglong * 16)()
        assert raw.asac_debug_td_stamps(st) == 0
        t = np.array(list(st)[:8], dtype=np.float64) / 100.0
        acc[1:] += np.diff(t)
        acc[0] += t[7] - t[0]
    names = ['whole workgroup', 'entry loads issued', 'temperature step (alpha_adam_block)', 'return terms -> LDS',
             'scan + TD error', 'election (NaN screen, pow, order test)', 'leaves + outputs stored', 'climb']
    for n, v in zip(names, acc / reps):
        print(f'{n:42s} {v:6.2f} us')


if __name__ == '__main__':
    main()
