#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one row per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []; cur = None
for line in txt.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    m = re.search(r'remark: +(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).split(' ')[0]] = int(m.group(2))
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, names):
    d = re.sub(r'^void ip::', '', d); d = re.sub(r'\(.*', '', d)
    print('%-90s vgpr=%-4s sgpr=%-4s scratch=%-5s occ=%s lds=%s' % (d[:90], r.get('VGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize'), r.get('Occupancy'), r.get('LDS')))
