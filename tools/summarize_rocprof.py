"""Compact a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names are truncated and
torch's template noise is stripped).  usage: summarize_rocprof.py stats.csv [steps] [top] [out.json]
The optional JSON holds, per library kernel (template arguments dropped, instances merged), the launch-weighted mean
duration and the launches per step: what bench.py's `*_in_situ` roofline fields are computed from."""
import csv
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', '', name)
    m = re.match(r'(void )?([A-Za-z_0-9:]+)', name)
    base = m.group(2) if m else name[:40]
    inner = re.search(r'(\w+Functor\w*|\w+_kernel_cuda|\w+_kernel_impl|\w+Ops|MeanOps|sum_functor|normal_kernel|uniform_kernel|FillFunctor|\w+_kernel)\b', name[len(base):])
    tag = inner.group(1) if inner else ''
    if name.startswith('Cijk'):
        return 'hipBLASLt ' + re.sub(r'_SN_.*', '', name)[:48]
    return (base + ('<' + tag + '>' if tag else ''))[:72]


def _lib_hash() -> str:
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location(
        'asac_build', Path(__file__).resolve().parent.parent / 'advanced-soft-actor-critic_amd' / 'csrc' / 'build.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f'# {path}: {len(rows)} kernels, {calls} launches, {tot / 1e6:.3f} ms total device time'
          f' ({calls / steps:.1f} launches/step, {tot / 1e3 / steps:.1f} us/step over {steps:g} steps)')
    print(f'{"kernel":72s} {"calls":>7s} {"/step":>6s} {"avg_us":>8s} {"min_us":>7s} {"tot%":>6s}')
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:top]:
        print(f'{short(r["Name"]):72s} {int(r["Calls"]):7d} {int(r["Calls"]) / steps:6.2f} '
              f'{float(r["AverageNs"]) / 1e3:8.2f} {float(r["MinNs"]) / 1e3:7.2f} {float(r["Percentage"]):6.2f}')
    asac = [r for r in rows if 'asac::' in r['Name']]
    if len(sys.argv) > 4:
        merged = {}
        for r in asac:
            k = re.match(r'(void )?(asac::[A-Za-z_0-9:]+)', r['Name']).group(2)
            m = merged.setdefault(k, [0, 0.0])
            m[0] += int(r['Calls'])
            m[1] += float(r['TotalDurationNs'])
        out = {k: {'avg_us': round(t / c / 1e3, 3), 'launches_per_step': round(c / steps, 3)} for k, (c, t) in merged.items()}
        out['_all'] = {'us_per_step': round(tot / 1e3 / steps, 1), 'launches_per_step': round(calls / steps, 1),
                       'asac_share': round(sum(float(r['TotalDurationNs']) for r in asac) / tot, 4), 'steps': steps}
        out['_meta'] = {'lib_hash': _lib_hash()}       # (bench.py quotes these durations only for this very library)
        json.dump(out, open(sys.argv[4], 'w'), indent=1, sort_keys=True)
    print(f'# asac kernels: {sum(float(r["TotalDurationNs"]) for r in asac) / tot * 100:.1f}% of device time, '
          f'{sum(int(r["Calls"]) for r in asac) / steps:.1f} launches/step')


if __name__ == '__main__':
    main()
