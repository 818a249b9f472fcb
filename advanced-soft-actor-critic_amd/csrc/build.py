"""Builds libasac_hip.so (gfx950) in-tree with hipcc.  `python build.py` or `build()`.

The library is kept next to the package (`advanced-soft-actor-critic_amd/lib/`), git-ignored but
shipped to the GPU box with the working tree.  hipcc cross-compiles without a GPU present.
"""
import hashlib
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
SOURCES = ['sumtree.hip', 'gather.hip', 'returns.hip', 'optim.hip', 'mlp.hip', 'gru.hip', 'gru_wide.hip', 'noise.hip', 'conv.hip', 'attn.hip', 'attn_mh.hip', 'rows_proj.hip', 'graph_fix.hip', 'xty.hip', 'linear.hip', 'episode.hip', 'decoder.hip', 'wide.hip', 'reduce.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError('hipcc not found (needed to build libasac_hip.so for gfx950)')


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode())
    for q in paths:
        h.update(q.name.encode())
        h.update(q.read_bytes())
    return h.hexdigest()


def _headers():
    return sorted(CSRC.glob('*.h')) + [REPO / 'include' / 'asac_hip.h']


def source_hash() -> str:
    """Content hash of everything libasac_hip.so is built from (every source, every header, the flags), 16 hex digits.
    Stamped into the committed profile summaries (`profiles/*_kernel_stats.json`, `*_pmc_traffic.json`: key `_meta`) by
    tools/summarize_rocprof.py / summarize_pmc.py; bench.py quotes a committed duration only when its stamp equals the
    hash of the tree it runs from."""
    return _digest([CSRC / s for s in SOURCES], [_digest(_headers(), FLAGS)])[:16]


def _stale_sources(obj_dir: Path):
    """Sources whose object is missing or was built from other text: the decision is by CONTENT (source + every header +
    flags, recorded beside the object), not by mtime — a checkout whose library happens to be newer than its sources is
    still rebuilt when they differ."""
    hdr = _digest(_headers(), FLAGS)
    stale = []
    for src in SOURCES:
        stem = Path(src).stem
        want = _digest([CSRC / src], [hdr])
        stamp = obj_dir / (stem + '.sha256')
        if not (obj_dir / (stem + '.o')).exists() or not stamp.exists() or stamp.read_text().strip() != want:
            stale.append((src, want))
    return stale


def needs_build() -> bool:
    return not LIB.exists() or bool(_stale_sources(LIB_DIR / 'obj'))


def build(force: bool = False, verbose: bool = True) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / 'obj'
    obj_dir.mkdir(exist_ok=True)
    stale = [(s, None) for s in SOURCES] if force else _stale_sources(obj_dir)
    if not stale and LIB.exists():
        return LIB
    hipcc = _hipcc()
    procs = []
    for src, _ in stale:
        obj = obj_dir / (Path(src).stem + '.o')
        (obj_dir / (Path(src).stem + '.sha256')).unlink(missing_ok=True)
        cmd = [hipcc, *FLAGS, f'-I{REPO / "include"}', f'-I{CSRC}', '-c', str(CSRC / src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd, src))
    for p, cmd, src in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    hdr = _digest(_headers(), FLAGS)
    for src, _ in stale:
        (obj_dir / (Path(src).stem + '.sha256')).write_text(_digest([CSRC / src], [hdr]))
    objs = [str(obj_dir / (Path(s).stem + '.o')) for s in SOURCES]
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', str(LIB)]
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
