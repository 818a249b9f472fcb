"""Per-kernel HBM traffic from rocprofv3 PMC passes.

usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json] [--by-grid]
(--by-grid keeps launches of one kernel with different grid sizes apart: size sweeps)

The two inputs come from SEPARATE passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir> -- python bench.py ...
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B derived from 64-B request counts.
gfx950 correction (MI355X_MICROARCH.md, "HBM"): a wide coalesced read is tallied at half its bytes, so
`fetch_bytes_corrected` = 2 x the raw figure; WRITE_SIZE is uncalibrated and reported raw.  Both are
per-launch averages over every dispatch of the kernel in the run.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r'\[clone.*', '', name).strip()
    m = re.match(r'(void )?(asac::[A-Za-z_0-9:]+)', name)
    if m:       # (nested namespaces kept: asac::dec::k_dec_fwd12)
        return m.group(2)
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', '', name)
    m = re.match(r'(void )?([A-Za-z_0-9:]+)', name)
    return (m.group(2) if m else name)[:60]


def load(path: str, counter: str, by_grid: bool) -> dict:
    per = defaultdict(list)
    with open(path) as f:
        rd = csv.DictReader(f)
        for row in rd:
            if row.get('Counter_Name') != counter:
                continue
            key = short(row['Kernel_Name'])
            if by_grid:
                key += f'@grid{row["Grid_Size"]}'
            per[key].append(float(row['Counter_Value']))
    return per


def main():
    by_grid = '--by-grid' in sys.argv
    sys.argv = [a for a in sys.argv if a != '--by-grid']
    fetch = load(sys.argv[1], 'FETCH_SIZE', by_grid)
    write = load(sys.argv[2], 'WRITE_SIZE', by_grid)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fr = 1024.0 * sum(f) / len(f) if f else None
        wr = 1024.0 * sum(w) / len(w) if w else None
        out[k] = {'launches': max(len(f), len(w)),
                  'fetch_bytes_raw': fr, 'fetch_bytes_corrected': None if fr is None else 2.0 * fr,
                  'write_bytes_raw': wr}
    print(f'{"kernel":48s} {"launches":>8s} {"fetch raw B":>13s} {"fetch x2 B":>13s} {"write raw B":>13s}')
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]['fetch_bytes_raw'] or 0)):
        fmt = lambda x: f'{x:13.0f}' if x is not None else f'{"-":>13s}'
        print(f'{k:48s} {v["launches"]:8d} {fmt(v["fetch_bytes_raw"])} {fmt(v["fetch_bytes_corrected"])} '
              f'{fmt(v["write_bytes_raw"])}')
    if len(sys.argv) > 3:
        from summarize_rocprof import _lib_hash
        out['_meta'] = {'lib_hash': _lib_hash()}       # (bench.py quotes this traffic only for this very library)
        json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
