"""gpurun_out/parity_errors.json (a run of the -m gpu tests under ASAC_PARITY_RECORD=1, i.e. under the call sites'
default bounds) -> tests/parity_tolerances.json (what the tests enforce: default x max(4 x used, 2 ulp), see
tests/parity_utils.py) and profiles/<round>_parity_errors.json (the observed errors, for DESIGN.md section 5).

    ASAC_PARITY_RECORD=1 python -m pytest tests -m gpu -q        (on the GPU box)
    python tools/set_tolerances.py [round] [--merge]

`--merge`: keep the table's other keys and, per key, the WORSE of the table's and the log's observed error (a partial run —
new tests only — extends the table instead of replacing it)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    argv = [a for a in sys.argv[1:] if a != '--merge']
    merge = '--merge' in sys.argv[1:]
    rnd = argv[0] if argv else 'r06'
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
    tol_path, rep_path = ROOT / 'tests' / 'parity_tolerances.json', ROOT / 'profiles' / f'{rnd}_parity_errors.json'
    if merge:
        old = json.loads(tol_path.read_text()) if tol_path.exists() else {}
        for key, rec in old.items():
            if key not in table or rec['used_of_default'] > table[key]['used_of_default']:
                table[key] = rec
        old_rep = json.loads(rep_path.read_text()) if rep_path.exists() else {}
        for key, rec in old_rep.items():
            if key not in report or rec.get('used_of_default', 0) > report[key].get('used_of_default', 0):
                report[key] = rec
    tol_path.write_text(json.dumps(table, indent=0, sort_keys=True))
    rep_path.write_text(json.dumps(report, indent=1, sort_keys=True))
    print(f'{len(table)} keys -> tests/parity_tolerances.json, profiles/{rnd}_parity_errors.json')
    worst = sorted(((v.get('enforced_rtol', 0), k) for k, v in report.items() if isinstance(v, dict) and 'enforced_rtol' in v), reverse=True)
    for rt, k in worst[:25]:
        print(f'  {k:60s} enforced rtol {rt:.2e}')


if __name__ == '__main__':
    main()
