"""The driver keeps a tail of bench.py's stdout and parses the LAST line: round 4's 21 KB record did not fit
(BENCH_r04 `parsed: null`).  The compact formatter must turn a full record — here the round-4 one, committed under
profiles/ — into a line under 4 KB that still carries the contract fields, one `roofline`, one `roofline_hbm` and
`cpu_baseline`."""
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _canned():
    return json.loads((ROOT / 'profiles' / 'r04_cfg2_bench.json').read_text().strip().splitlines()[-1])


def test_compact_line_fits_and_round_trips():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 20000            # the record that broke the driver's parser
    line = bench.compact_line(full)
    assert '\n' not in line and len(line.encode()) < 4096
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d['value'] == full['value'] and d['ms_per_step'] == full['ms_per_step']
    assert set(d['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(d['roofline_hbm']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(d['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'}
    assert d['config']['workload'] and 'model' not in d['config']
    assert all(len(v) <= 160 for v in d['config'].values() if isinstance(v, str))
    assert set(d['side_configs']) == set(full['configs'])


def test_compact_line_worst_case_still_parses():
    """long strings everywhere, many side configurations: the optional parts are shed before the headline is"""
    import bench
    full = _canned()
    full['config']['workload'] = 'w' * 5000
    full['roofline']['kernel'] = 'k' * 5000
    full['cpu_baseline']['sample'] = 's' * 5000
    full['configs'] = {f'cfg_{i}': {'value': 1.0 * i, 'roofline_hbm': {'frac': 0.1}} for i in range(400)}
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096
    d = json.loads(line)
    assert d['value'] == full['value'] and d['roofline'] is not None and d['cpu_baseline'] is not None


def test_compact_line_without_optional_sections():
    import bench
    full = _canned()
    for k in ('roofline', 'roofline_hbm', 'cpu_baseline', 'configs', 'sweep', 'kernels'):
        full[k] = None
    d = json.loads(bench.compact_line(full))
    assert d['roofline'] is None and d['cpu_baseline'] is None and 'side_configs' not in d


def test_every_sample_return_entry_point_has_its_kernel_and_bytes():
    """a new entry point of the K1-K4 group that bench.py does not know silently drops out of `roofline_hbm` (round 5: the
    gather launch with the IS weights was missing for one refresh, K1-K4 read 0.15 % instead of 33 %)"""
    import bench
    assert all(n in bench._KERNEL_OF for n in bench.SAMPLE_RETURN)
    import re
    from pathlib import Path
    header = (Path(bench.__file__).parent / 'include' / 'asac_hip.h').read_text()
    declared = set(re.findall(r'\bint (asac_(?:step_prologue_sample\w*|window_gather_pad\w*|sumtree_sample|vtrace_return_min|td_update))\(', header))
    declared -= {'asac_vtrace_return_min_sc', 'asac_window_gather_plan'}       # (bound through their base names / not a launch)
    assert declared <= set(bench.SAMPLE_RETURN), declared - set(bench.SAMPLE_RETURN)
