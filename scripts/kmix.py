"""CPU: instruction mix between consecutive s_barrier of one kernel in a -save-temps assembly file.
python scripts/kmix.py <file.s> <kernel-name-substring>"""
import collections, re, sys
txt = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(txt) if re.match(r'^_Z\w*%s\w*:' % re.escape(sys.argv[2]), l))
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
L = txt[start:end]
def cls(l):
    t = l.strip().split()[0] if l.strip() else ''
    if t.startswith('v_mfma'): return 'mfma'
    if t.startswith('v_'): return 'valu'
    if t.startswith('ds_'): return t
    if t.startswith('buffer_') or t.startswith('global_'): return t
    if t.startswith('s_nop'): return 's_nop'
    if t.startswith('s_waitcnt'): return 'waitcnt'
    if t.startswith('s_cbranch') or t.startswith('s_branch'): return 'branch'
    if t.startswith('s_'): return 'salu'
    return None
idx = [i for i, l in enumerate(L) if 's_barrier' in l]
print(len(L), 'lines; barriers at', idx)
for a, b in zip([0] + idx, idx + [len(L)]):
    c = collections.Counter(cls(l) for l in L[a:b] if cls(l))
    print(a, b, dict(c))
