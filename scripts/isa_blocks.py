"""Instruction-class counts per basic block of one kernel in a hipcc --save-temps .s file (epilogue VALU accounting).
   python scripts/isa_blocks.py conv-hip-amdgcn-amd-amdhsa-gfx950.s 'conv3x3_sp_kernelILi128ELb0ELi0ELi0ELb0ELb0E' [min]"""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = re.search(r'^(_ZN\S*' + re.escape(key) + r'\S*):', s, re.M)
start = m.end()
end = s.index('.Lfunc_end', start)
blocks, cur = [], ['entry', collections.Counter()]
for l in s[start:end].split('\n'):
    t = l.strip()
    if re.match(r'^\.LBB\S+:', t):
        blocks.append(cur)
        cur = [t.split(':')[0], collections.Counter()]
        continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    cls = ('mfma' if 'mfma' in op else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else
           'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_', 'scratch_')) else 'other')
    cur[1][cls] += 1
    if cls == 'valu':
        cur[1]['  ' + op] += 1
blocks.append(cur)
print(m.group(1))
for name, c in blocks:
    tot = sum(v for k, v in c.items() if not k.startswith('  '))
    if tot >= minc:
        print(name, {k: v for k, v in c.items() if not k.startswith('  ')})
        ops = sorted(((v, k.strip()) for k, v in c.items() if k.startswith('  ')), reverse=True)[:12]
        print('      ', ', '.join(f'{k} {v}' for v, k in ops))
