"""Per-kernel SQ counter summary from one rocprofv3 PMC pass (the latency picture of the step's kernels).

usage: summarize_sq.py <counter_collection.csv>
pass:  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
           SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d <dir> -- python bench.py ...
MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY (wave parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall)
+ ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves); SQ_VALU_MFMA_BUSY_CYCLES counts cycles the MFMA pipe
is busy (summed over SIMDs), SQ_BUSY_CYCLES cycles any wave is resident (per SE).  Per-launch averages.
"""
import csv
import sys
from collections import defaultdict

from summarize_pmc import short

NAMES = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS',
         'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT']


def main():
    per = defaultdict(lambda: defaultdict(list))
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            per[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    print('# per-launch averages; wave cycles in quad-cycles summed over the launch\'s waves; parked / stalled / issuing = shares of them')
    print(f'{"kernel":36s} {"launches":>8s} {"wave cyc":>10s} {"parked":>7s} {"stalled":>8s} {"issuing":>8s} {"lds-stall":>9s} '
          f'{"mfma busy cyc":>13s} {"mfma/wave-cyc":>13s} {"lds conflict":>12s}')
    rows = []
    for k, c in per.items():
        if not k.startswith('asac::'):
            continue
        avg = {n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0) for n in NAMES}
        rows.append((k, len(c.get('SQ_WAVE_CYCLES', [])), avg))
    for k, n, a in sorted(rows, key=lambda r: -r[2]['SQ_WAVE_CYCLES'] * r[1]):
        w = a['SQ_WAVE_CYCLES'] or 1.0
        print(f'{k:36s} {n:8d} {a["SQ_WAVE_CYCLES"]:10.0f} {a["SQ_WAIT_ANY"] / w:7.1%} {a["SQ_WAIT_INST_ANY"] / w:8.1%} '
              f'{a["SQ_ACTIVE_INST_ANY"] / w:8.1%} {a["SQ_WAIT_INST_LDS"] / w:9.1%} {a["SQ_VALU_MFMA_BUSY_CYCLES"]:13.0f} '
              f'{a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * w):13.2%} {a["SQ_LDS_BANK_CONFLICT"]:12.0f}')


if __name__ == '__main__':
    main()
