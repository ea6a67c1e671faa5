"""compressed view of a kernel's load / MFMA / wait schedule from hipcc -S output: tools/isa_schedule.py file.s kernel_substring"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
pat = re.compile(r'\s*(global_load_\w+|global_store_\w+|v_mfma_\w+|s_waitcnt|s_barrier|s_cbranch_\w+|ds_read\w*|ds_write\w*|buffer_\w+|scratch_\w+|global_atomic\w+)')
seq = []
for l in txt[start:end]:
    m = pat.match(l)
    if m:
        k = m.group(1)
        if k == 's_waitcnt':
            k = 'wait ' + l.split('s_waitcnt', 1)[1].split(';')[0].strip()
        seq.append(k)
    elif re.match(r'^\.LBB', l):
        seq.append('--- ' + l.strip())
out, prev, c = [], None, 0
for k in seq:
    if k == prev:
        c += 1
    else:
        if prev:
            out.append(f'{c:4d} x {prev}')
        prev, c = k, 1
out.append(f'{c:4d} x {prev}')
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 400
print('\n'.join(out[:limit]))
