"""Print the kernel sequence of a typical train step (the one of median span among the last 50) found in a
rocprofv3 `*_kernel_trace.csv`
(one line per dispatch: start offset, duration, gap to the previous kernel's end, short name).
usage: step_sequence.py kernel_trace.csv [anchor-kernel-substring]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r'at::native::(\(anonymous namespace\)::)?', '', name)
    m = re.match(r'(void )?([A-Za-z_0-9:]+)', name)
    base = m.group(2) if m else name[:40]
    inner = re.search(r'(\w+Functor\w*|\w+_kernel_cuda|\w+_kernel_impl|MeanOps|sum_functor|normal_kernel|uniform_kernel|FillFunctor)', name[len(base):])
    return (base + ('<' + inner.group(1) + '>' if inner else ''))[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a step begins with its sampler: the fused prologue + sample launch, or the stand-alone sampler
    anchors = [sys.argv[2]] if len(sys.argv) > 2 else ['k_prologue_sample', 'k_sumtree_sample']
    idx = []
    for anchor in anchors:
        idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
        if len(idx) >= 3:
            break
    if len(idx) < 3:
        print('anchor not found often enough')
        return
    # a full step in steady state: the one of median span among the last 50 (a single step can contain a host or
    # profiler stall of milliseconds)
    cand = list(zip(idx[-52:-2], idx[-51:-1])) or [(idx[-3], idx[-2])]
    spans = sorted((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']), a, b) for a, b in cand)
    _, lo, hi = spans[len(spans) // 2]
    t0 = int(rows[lo]['Start_Timestamp'])
    prev_end = None
    busy = 0
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0 if prev_end is None else s - prev_end
        busy += e - s
        print(f'{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {gap / 1e3:7.2f}  {short(r["Kernel_Name"])}')
        prev_end = e
    span = int(rows[hi]['Start_Timestamp']) - t0
    print(f'# {hi - lo} launches, span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us')


if __name__ == '__main__':
    main()
