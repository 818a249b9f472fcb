"""`-Rpass-analysis=kernel-resource-usage` of every kernel of libasac_hip.so (hipcc cross-compiles without a GPU):
registers, spills, scratch, occupancy -> profiles/<round>_kernel_resource_usage.txt.   usage: resource_usage.py [round]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / 'advanced-soft-actor-critic_amd' / 'csrc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '--cuda-device-only',
         '-Rpass-analysis=kernel-resource-usage', f'-I{ROOT / "include"}', f'-I{CSRC}']


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    rows = []
    for src in sorted(CSRC.glob('*.hip')):
        r = subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, '-c', str(src), '-o', '/dev/null'], capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = {'kernel': re.sub(r'\(.*', '', name), 'file': src.name}
                rows.append(cur)
                continue
            m = re.search(r'remark:\s+([\w \[\]/]+?): (\S+) \[-Rpass', line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    out = [f'# hipcc {" ".join(FLAGS[:4])} -Rpass-analysis=kernel-resource-usage, every kernel of libasac_hip.so',
           f'{"kernel":78s} {"file":12s} {"SGPR":>5s} {"VGPR":>5s} {"AGPR":>5s} {"scratch B/lane":>14s} {"occupancy":>9s} '
           f'{"SGPR spills":>11s} {"VGPR spills":>11s}']
    for r in rows:
        out.append(f'{r["kernel"][:78]:78s} {r["file"]:12s} {r.get("TotalSGPRs", "?"):>5s} {r.get("VGPRs", "?"):>5s} '
                   f'{r.get("AGPRs", "?"):>5s} {r.get("ScratchSize [bytes/lane]", "?"):>14s} '
                   f'{r.get("Occupancy [waves/SIMD]", "?"):>9s} {r.get("SGPRs Spill", "?"):>11s} {r.get("VGPRs Spill", "?"):>11s}')
    (ROOT / 'profiles' / f'{rnd}_kernel_resource_usage.txt').write_text('\n'.join(out) + '\n')
    print('\n'.join(o for o in out if 'k_pi_sample_q' in o or 'k_policy_step' in o or o.startswith(('#', 'kernel'))))


if __name__ == '__main__':
    main()
