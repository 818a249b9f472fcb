"""Scan the ISA of every kernel of the library for chains `load, wait, load, wait, ...` (a `global_load` followed by
`s_waitcnt vmcnt(0|1)` before the next one): candidates for loads that were meant as one batch but serialise — a load under a
lane condition becomes a branch with its own wait (NOTES.md §1 round 5: `k_dec_bwd3`).  Many hits are genuine dependencies
(election chains, pointer-table reads): read the code before changing it, and measure — the rewrite cost registers in k_attn_mh.
usage: python tools/debug/serial_loads.py        (compiles csrc/*.hip to /tmp/asac_asm/*.s with hipcc, gfx950)"""
import glob
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CSRC = ROOT / 'advanced-soft-actor-critic_amd' / 'csrc'
OUT = Path('/tmp/asac_asm')
OUT.mkdir(exist_ok=True)
procs = []
for src in sorted(CSRC.glob('*.hip')):
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                                   f'-I{ROOT / "include"}', f'-I{CSRC}', '-S', '--cuda-device-only', str(src), '-o',
                                   str(OUT / (src.stem + '.s'))], stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
for path in sorted(glob.glob(str(OUT / '*.s'))):
    rows, name = [], None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, seq = m.group(1), []
            rows.append((name, seq))
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith('global_load') or t.startswith('buffer_load'):
            seq.append('L')
        elif t.startswith('s_waitcnt') and ('vmcnt(0)' in t or 'vmcnt(1)' in t):
            seq.append('W')
    for name, seq in rows:
        runs = re.findall(r'(?:L{1,2}W){4,}', ''.join(seq))
        if runs:
            print(f'{Path(path).name:14s} {name[:72]:72s} serialised groups: {[r.count("W") for r in runs]}')
