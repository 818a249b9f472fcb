"""CPU: the C-ABI library loads and exports exactly what include/asac_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / 'include' / 'asac_hip.h'


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(asac_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert 'asac_sumtree_sample' in syms and 'asac_window_gather_pad' in syms and len(syms) >= 18


def test_library_exports_every_declared_symbol():
    from asac_amd import native
    if not native.LIB_PATH.exists():
        import importlib.util
        spec = importlib.util.spec_from_file_location('asac_build', ROOT / 'advanced-soft-actor-critic_amd/csrc/build.py')
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    lib = ctypes.CDLL(str(native.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in asac_hip.h but not exported'


def test_integration_notes_name_every_entry_point():
    """INTEGRATION.md maps every export to the reference lines it replaces: a new entry point without a row there is
    a boundary change nobody wrote down."""
    notes = (ROOT / 'INTEGRATION.md').read_text()
    missing = [name for name in declared_symbols() if name not in notes]
    assert not missing, missing


def test_binding_covers_header_and_abi_version():
    from asac_amd import native
    assert sorted(native.EXPORTED_SYMBOLS) == declared_symbols()
    lib = native.load()
    m = re.search(r'#define ASAC_ABI_VERSION (\d+)', HEADER.read_text())
    assert lib.asac_version() == int(m.group(1)) == native.ABI_VERSION


def test_struct_layouts_match_header_sizes():
    """ctypes mirrors of the by-value structs against sizeof as the library's compiler laid them out"""
    from asac_amd import native
    lib = native.load()
    mirrors = {'asac_gather_key_t': native.GatherKey, 'asac_row_move_t': native.RowMove, 'asac_sidecar_t': native.Sidecar,
               'asac_squash_job_t': native.SquashJob, 'asac_vtrace_args_t': native.VtraceArgs,
               'asac_mlp_desc_t': native.MlpDesc, 'asac_mlp_job_t': native.MlpJob, 'asac_pi_q_job_t': native.PiQJob, 'asac_mlp_sample_epilogue_t': native.SampleEpilogue,
               'asac_gru_desc_t': native.GruDesc, 'asac_conv2_desc_t': native.Conv2Desc, 'asac_partial_sum_t': native.PartialSum}
    for name, mirror in mirrors.items():
        assert lib.asac_struct_size(name.encode()) == ctypes.sizeof(mirror), name
    assert lib.asac_struct_size(b'no_such_struct') == -1


def test_product_path_refuses_cpu_devices():
    import torch
    from asac_amd import native
    from algorithm.replay_buffer import PrioritizedReplayBuffer
    with pytest.raises(native.AsacNativeError):
        PrioritizedReplayBuffer(batch_size=4, device=torch.device('cpu'), capacity=16)
    from algorithm.agent import EpisodeSlab
    import numpy as np
    with pytest.raises(native.AsacNativeError):     # the agent side's episode slabs live in HBM as well
        EpisodeSlab([(6,)], [np.float32], 2, (0,), 16, torch.device('cpu'), np.zeros(2, np.float32))


def test_binding_argument_counts_match_the_header():
    """every ctypes signature lists as many arguments as the header declares (ctypes accepts EXTRA arguments silently and
    converts them by its default rules — a Python int would travel as a 32-bit int where the C side expects int64)"""
    import re
    from asac_amd import native
    hdr = re.sub(r'/\*.*?\*/', '', (ROOT / 'include' / 'asac_hip.h').read_text(), flags=re.S)
    bad = []
    for name, (_res, args) in native._SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;{]*?)\)\s*;', hdr, re.S)
        assert m, f'{name}: bound but not declared'
        params = [p_ for p_ in m.group(1).split(',') if p_.strip() and p_.strip() != 'void']
        if len(params) != len(args):
            bad.append((name, len(params), len(args)))
    assert not bad, bad
