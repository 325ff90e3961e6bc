#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: kernel (demangled), VGPRs, AGPRs, scratch, LDS, occupancy.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; tools/kres.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    d = re.sub(r'\(.*', '', d).replace('upf::', '')
    if flt and flt not in d:
        continue
    print('%-70s vgpr %3d agpr %3d scratch %4d lds %6d occ %d' % (d[:70], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'LDS Size \[bytes/block\]'), g(r'Occupancy \[waves/SIMD\]')))
