#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '-std=c++17', '-O3', '--offload-arch=gfx950', '-fno-gpu-rdc', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'remark: .*?Function Name: (\S+)', line) or re.search(r'Name: (\S+)', line)
    if m and 'Function Name' in line or (m and ' Name:' in line):
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()[:90]}
        rows.append(cur)
        continue
    for key in ['VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]', 'VGPR Spill', 'SGPR Spill']:
        m = re.search(re.escape(key) + r': (\d+)', line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
for r in rows:
    if flt and flt not in r['name']:
        continue
    print(f"{r['name']:<92} V={r.get('VGPRs')} S={r.get('TotalSGPRs')} scr={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
