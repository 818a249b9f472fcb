"""prints the headline numbers of a bench.py JSON line read from stdin"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['config'].get('workload', '')[:60])
for k, v in (d.get('kernels') or {}).items():
    print(f"  {k:40s} {v['launches_per_step']:5.2f} x {v['med_us']:7.2f} us (avg {v['avg_us']:.2f})")
