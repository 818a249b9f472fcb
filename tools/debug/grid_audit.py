"""Per kernel of a rocprofv3 kernel trace: workgroups per launch, LDS per workgroup, how many workgroups fit a CU by LDS, and
how many rounds of resident workgroups the launch takes on 256 CUs (a launch of 288 one-per-CU workgroups is two rounds, the
second with 32): python tools/debug/grid_audit.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0, 0, 0, 0.0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r['Kernel_Name'].split('(')[0][:48]
        wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        lds = int(r.get('LDS_Block_Size', r.get('LDS_Block_Size_In_Bytes', 0)) or 0)
        vgpr = int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0)
        key = (name, grid // max(wg, 1), wg, lds, vgpr)
        e = rows[key]
        e[4] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e[5] += 1
print(f'{"kernel":48s} {"WGs":>6s} {"thr":>5s} {"LDS KB":>7s} {"regs":>5s} {"fit/CU":>6s} {"rounds":>7s} {"avg us":>8s} {"calls":>6s}')
for (name, wgs, wg, lds, vgpr), e in sorted(rows.items(), key=lambda kv: -kv[1][4]):
    waves = max(1, wg // 64)
    by_lds = 160 * 1024 // lds if lds else 32
    by_reg = (512 // max(vgpr, 1)) * 4 // waves if vgpr else 32
    fit = max(1, min(by_lds, by_reg, 32 // waves))
    print(f'{name:48s} {wgs:6d} {wg:5d} {lds / 1024:7.1f} {vgpr:5d} {fit:6d} {wgs / (256 * fit):7.2f} {e[4] / e[5]:8.1f} {e[5]:6d}')
