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
    assert set(d['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    assert d['config']['workload'] and 'model' not in d['config']
    assert all(len(v) <= bench.PROSE_LIMIT for v in d['config'].values() if isinstance(v, str))
    assert len(d['cpu_baseline']['sample']) <= bench.PROSE_LIMIT
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


def test_sizes_and_core_counts_travel_as_numbers():
    """round 5: the driver's copy of the line ended "host cpu_count=2" (it was 256) and "262144 transit" — prose was clipped
    mid-number.  Core counts, thread counts and resident rows are numeric fields; prose stays under PROSE_LIMIT."""
    import bench
    full = _canned()
    full['config'].update(rows_resident=262144, replay_capacity=524288, n_step=4, burn_in_step=0,
                          workload='cfg2: TEST vector obs(6) c_action(2) stock MLP, PER capacity 524288, n_step=4 V-trace')
    full['cpu_baseline'].update(threads=1, host_cores=256, rows_resident=262144, sample_steps=1234, sample_seconds=9.5,
                                sample='1234 steps of the same workload in 9.5s, oracle/sac_ref.SacRef, best of a thread sweep')
    d = json.loads(bench.compact_line(full))
    assert d['config']['rows_resident'] == 262144 and d['config']['replay_capacity'] == 524288
    assert d['cpu_baseline']['host_cores'] == 256 and d['cpu_baseline']['threads'] == 1
    assert d['cpu_baseline']['rows_resident'] == 262144
    assert d['config']['workload'] == full['config']['workload']                    # (fits: not clipped)
    assert d['cpu_baseline']['sample'] == full['cpu_baseline']['sample']
    for text in (d['config']['workload'], d['cpu_baseline']['sample']):
        assert len(text) <= bench.PROSE_LIMIT


def _live_roofline():
    return {'bound': 'mfma', 'achieved': 6.78, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 0.0431, 'traffic': None,
            'alg_flops_per_launch': 76480512, 'kernel': 'asac::k_pi_sample_q', 'avg_launch_us': 11.28}


def test_committed_duration_is_quoted_only_for_the_running_library(tmp_path):
    """`roofline.frac` comes from the committed rocprofv3 summary ONLY when that summary carries the content hash of the
    library that is running; any other stamp (a kernel changed since the profile was taken), a missing stamp or a missing
    file gives the LIVE HIP-event figure, labelled as such — never a stale duration beside live work counts."""
    import bench
    stats = {'asac::k_pi_sample_q': {'avg_us': 10.0, 'launches_per_step': 2.0}, '_meta': {'lib_hash': 'abc'}}
    pmc = {'asac::k_pi_sample_q<1>': {'launches': 10, 'fetch_bytes_corrected': 1000.0, 'write_bytes_raw': 24.0}, '_meta': {'lib_hash': 'abc'}}
    (tmp_path / f'{bench.ROUND}_cfgx_kernel_stats.json').write_text(json.dumps(stats))
    (tmp_path / f'{bench.ROUND}_cfgx_pmc_traffic.json').write_text(json.dumps(pmc))
    live = _live_roofline()
    r = bench.roofline_with_profile(live, 'cfgx', 'asac::k_pi_sample_q', tmp_path, 'abc')
    assert r['frac_source'] == f'profiles/{bench.ROUND}_cfgx_kernel_stats.json'
    assert r['avg_launch_us'] == 10.0 and r['achieved'] == pytest.approx(7.648, abs=1e-3)
    assert r['frac'] == pytest.approx(7.648 / 157.3, abs=1e-5) and r['traffic'] == 1024
    assert r['frac_hip_events'] == live['frac'] and r['avg_launch_us_hip_events'] == live['avg_launch_us']
    # another library: live figures, said so
    r = bench.roofline_with_profile(live, 'cfgx', 'asac::k_pi_sample_q', tmp_path, 'other')
    assert r['frac_source'] == 'live (profile stale)' and r['frac'] == live['frac'] and r['achieved'] == live['achieved']
    assert r['avg_launch_us'] == live['avg_launch_us'] and r['traffic'] is None
    # no stamp at all (summaries of earlier rounds): stale too
    del stats['_meta']
    (tmp_path / f'{bench.ROUND}_cfgx_kernel_stats.json').write_text(json.dumps(stats))
    assert bench.roofline_with_profile(live, 'cfgx', 'asac::k_pi_sample_q', tmp_path, 'abc')['frac_source'] == 'live (profile stale)'
    # no file
    r = bench.roofline_with_profile(live, 'cfgy', 'asac::k_pi_sample_q', tmp_path, 'abc')
    assert r['frac_source'] == 'live (no committed profile)' and r['frac'] == live['frac']
    # and the label survives the compact line
    full = _canned()
    full['roofline'] = r
    assert json.loads(bench.compact_line(full))['roofline']['frac_source'] == 'live (no committed profile)'


def test_no_fallback_to_older_rounds():
    import bench
    assert not hasattr(bench, '_ROUNDS')
    # the committed summaries of THIS round (if any yet) carry a stamp
    for path in (ROOT / 'profiles').glob(f'{bench.ROUND}_*_kernel_stats.json'):
        assert (json.loads(path.read_text()).get('_meta') or {}).get('lib_hash'), path.name


def test_library_hash_follows_the_sources(tmp_path):
    import bench
    h = bench.library_hash()
    assert len(h) == 16 and h == bench.library_hash()
