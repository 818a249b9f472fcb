"""Copy the artifacts `tools/refresh_profiles.sh` left in gpurun_out/refresh/ into profiles/ (round-tagged
names, PMC summaries reduced to the library's own kernels) and print the headline numbers."""
import json
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
R, P, TAG = ROOT / 'gpurun_out' / 'refresh', ROOT / 'profiles', 'r06'

CFGS = ('cfg2', 'cfg3', 'cfg4', 'cfg4_84', 'cfg5')
for c in CFGS:
    # (one JSON document per file: cfg2's is bench_details.json, indented; the others the `--emit full` line)
    d = json.loads((R / f'{c}_bench.json').read_text().strip().splitlines()[-1] if c != 'cfg2' else (R / f'{c}_bench.json').read_text())
    (P / f'{TAG}_{c}_bench.json').write_text(json.dumps(d) + '\n')
shutil.copy(R / 'cfg2_bench_line.json', P / f'{TAG}_cfg2_bench_line.json')
for name in ('cfg2_kernel_stats.csv', 'cfg2_kernel_stats_summary.txt', 'kernel_sweep.txt', 'cfg2_step_sequence.txt',
             'cfg3_step_sequence.txt', 'cfg4_step_sequence.txt', 'cfg5_step_sequence.txt', 'cfg3_kernel_stats_summary.txt',
             'cfg4_kernel_stats_summary.txt', 'cfg5_kernel_stats_summary.txt', 'cfg5_without_prediction_step_sequence.txt',
             'cfg5_without_prediction_kernel_stats_summary.txt', 'cfg2_kernel_stats.json', 'cfg3_kernel_stats.json',
             'cfg4_kernel_stats.json', 'cfg5_kernel_stats.json', 'cfg4_84_kernel_stats.json', 'cfg4_84_kernel_stats_summary.txt', 'cfg4_84_step_sequence.txt', 'cfg5_without_prediction_kernel_stats.json',
             'cfg3_h64_kernel_stats.json', 'cfg3_h64_kernel_stats_summary.txt', 'cfg3_h64_step_sequence.txt',
             'cfg_attn_h64_kernel_stats.json', 'cfg_attn_h64_kernel_stats_summary.txt', 'cfg_attn_h64_step_sequence.txt',
             'cfg2_lookahead_step_sequence.txt', 'cfg2_lookahead_kernel_stats.json', 'cfg2_lookahead_kernel_stats_summary.txt'):
    if (R / name).exists():
        shutil.copy(R / name, P / f'{TAG}_{name}')
for src, dst, title in (('kernel_sweep_pmc.json', f'{TAG}_kernel_sweep_pmc',
                         'tools/kernel_sweep.py under rocprofv3 --pmc (two passes: FETCH_SIZE, WRITE_SIZE); per-launch averages, keyed kernel@grid'),
                        ('cfg2_pmc_traffic.json', f'{TAG}_cfg2_pmc_traffic',
                         'bench.py cfg2 (--steps 100 --warmup 10 --fill 20000, hipgraph replay) under rocprofv3 --pmc (two passes); per-launch averages'),
                        *[(f'{c}_pmc_traffic.json', f'{TAG}_{c}_pmc_traffic',
                           f'bench.py --config {c} (--steps 60 --warmup 10 --fill 20000, hipgraph replay) under rocprofv3 --pmc (two passes); per-launch averages')
                          for c in ('cfg3', 'cfg3_h64', 'cfg4', 'cfg4_84', 'cfg5', 'cfg_attn_h64') if (R / f'{c}_pmc_traffic.json').exists()]):
    raw = json.load(open(R / src))
    d = {k: v for k, v in raw.items() if k.startswith('asac::')}
    json.dump({**d, **({'_meta': raw['_meta']} if '_meta' in raw else {})}, open(P / f'{dst}.json', 'w'), indent=1, sort_keys=True)
    lines = [f'# {title}', '# fetch x2 = gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md "HBM"); write is raw',
             f'{"kernel":52s} {"launches":>8s} {"fetch raw B":>13s} {"fetch x2 B":>13s} {"write raw B":>13s}']
    fmt = lambda x: f'{x:13.0f}' if x is not None else f'{"-":>13s}'
    for k, v in sorted(d.items(), key=lambda kv: -(kv[1]['fetch_bytes_raw'] or 0)):
        lines.append(f'{k:52s} {v["launches"]:8d} {fmt(v["fetch_bytes_raw"])} {fmt(v["fetch_bytes_corrected"])} {fmt(v["write_bytes_raw"])}')
    (P / f'{dst}.txt').write_text('\n'.join(lines) + '\n')
for c in CFGS:
    d = json.load(open(P / f'{TAG}_{c}_bench.json'))
    r, h = d['roofline'] or {}, d['roofline_hbm'] or {}
    print(c, d['value'], d['ms_per_step'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], '| roofline',
          r.get('kernel'), r.get('achieved'), r.get('unit'), r.get('frac'), r.get('traffic'), '| hbm', h.get('kernels') or h.get('kernel'),
          h.get('achieved'), h.get('frac'))
