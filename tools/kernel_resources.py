#!/usr/bin/env python3
"""VGPR / AGPR / LDS / occupancy of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
cur, rows = None, {}
for ln in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur).replace('void cmoop::', '')
        rows[cur] = {}
    for k in ('VGPRs', 'AGPRs', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]', 'ScratchSize [bytes/lane]'):
        m = re.search(r' ' + re.escape(k) + r': (\d+)', ln)
        if m and cur:
            rows[cur][k] = int(m.group(1))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, v in rows.items():
    if flt in k:
        print(f"{k:50s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):4d} occ {v.get('Occupancy [waves/SIMD]')} LDS {v.get('LDS Size [bytes/block]')} scratch {v.get('ScratchSize [bytes/lane]')}")
