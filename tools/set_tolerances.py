"""gpurun_out/parity_errors.json (a run of the -m gpu tests under ASAC_PARITY_RECORD=1, i.e. under the call sites'
default bounds) -> tests/parity_tolerances.json (what the tests enforce: default x max(4 x used, 2 ulp), see
tests/parity_utils.py) and profiles/<round>_parity_errors.json (the observed errors, for DESIGN.md section 5).

    ASAC_PARITY_RECORD=1 python -m pytest tests -m gpu -q        (on the GPU box)
    python tools/set_tolerances.py [round]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    log = json.loads((ROOT / 'gpurun_out' / 'parity_errors.json').read_text())
    table, report = {}, {}
    for key, rec in sorted(log.items()):
        if 'used_of_default' not in rec:
            report[key] = rec
            continue
        used = rec['used_of_default']
        rt = rec['default_rtol']
        scale = min(1., max(4. * used, 2.4e-7 / max(rt, 1e-300)))
        table[key] = {'used_of_default': used}
        slim = {k: rec[k] for k in ('max_abs', 'max_rel', 'max_err_over_tensor_max', 'strict_max_abs', 'slack_max_over_lr_steps',
                                    'slack_entries', 'entries', 'calls', 'tensors') if k in rec}
        atol_key = 'default_atol' if 'default_atol' in rec else 'default_atol_frac'
        slim.update(default_rtol=rt, **{atol_key: rec[atol_key]}, used_of_default=used,
                    enforced_rtol=rt * scale, **{'enforced_' + atol_key[8:]: rec[atol_key] * scale})
        report[key] = slim
    (ROOT / 'tests' / 'parity_tolerances.json').write_text(json.dumps(table, indent=0, sort_keys=True))
    (ROOT / 'profiles' / f'{rnd}_parity_errors.json').write_text(json.dumps(report, indent=1, sort_keys=True))
    print(f'{len(table)} keys -> tests/parity_tolerances.json, profiles/{rnd}_parity_errors.json')
    worst = sorted(((v.get('enforced_rtol', 0), k) for k, v in report.items() if isinstance(v, dict) and 'enforced_rtol' in v), reverse=True)
    for rt, k in worst[:25]:
        print(f'  {k:60s} enforced rtol {rt:.2e}')


if __name__ == '__main__':
    main()
