"""Instruction histogram of one kernel from hipcc's -S output.  usage: isa_hist.py file.s mangled_name_prefix"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n'); name = sys.argv[2]
i = [k for k, l in enumerate(L) if l.startswith(name) and ':' in l.split(';')[0]][0]
h = collections.Counter(); n = 0
for t in (x.strip() for x in L[i + 1:]):
    if t.startswith('s_endpgm'): break
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    h[op] += 1; n += 1
cls = collections.Counter()
for op, c in h.items():
    k = 'valu' if op.startswith('v_') and not op.startswith('v_mfma') else 'mfma' if op.startswith('v_mfma') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'
    cls[k] += c
print(n, dict(cls))
for op, c in h.most_common(40): print(f"  {op:32s} {c}")
