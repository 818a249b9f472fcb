"""Builds libasac_hip.so (gfx950) in-tree with hipcc.  `python build.py` or `build()`.

The library is kept next to the package (`advanced-soft-actor-critic_amd/lib/`), git-ignored but
shipped to the GPU box with the working tree.  hipcc cross-compiles without a GPU present.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
REPO = PKG.parent
LIB_DIR = PKG / 'lib'
LIB = LIB_DIR / 'libasac_hip.so'
SOURCES = ['sumtree.hip', 'gather.hip', 'returns.hip', 'optim.hip', 'mlp.hip', 'gru.hip', 'gru_wide.hip', 'noise.hip', 'conv.hip', 'attn.hip', 'attn_mh.hip', 'graph_fix.hip', 'xty.hip', 'linear.hip', 'episode.hip', 'decoder.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError('hipcc not found (needed to build libasac_hip.so for gfx950)')


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + sorted(CSRC.glob('*.h')) + [REPO / 'include' / 'asac_hip.h']
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / 'obj'
    obj_dir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = obj_dir / (Path(src).stem + '.o')
        cmd = [hipcc, *FLAGS, f'-I{REPO / "include"}', f'-I{CSRC}', '-c', str(CSRC / src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(str(obj))
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', str(LIB)]
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
