"""Prints the memory / MFMA / wait sequence of one kernel from hipcc's -S output (finds compiler-inserted drains).
usage: isa_seq.py file.s mangled_name_prefix
D LDS-DMA, L other load, Wn s_waitcnt vmcnt(n), S store, M mfma, r ds_read, w ds_write, B barrier, | branch"""
import re, sys
L = open(sys.argv[1]).read().split('\n'); name = sys.argv[2]
i = [k for k, l in enumerate(L) if l.startswith(name) and ':' in l.split(';')[0]][0]; seq = []
for t in (x.strip() for x in L[i + 1:]):
    if t.startswith('s_endpgm'): break
    op = t.split()[0] if t else ''
    if op.startswith('buffer_load') and ' lds' in t: seq.append('D')
    elif op.startswith('buffer_load') or op.startswith('global_load'): seq.append('L')
    elif op == 's_waitcnt' and 'vmcnt' in t: seq.append('W' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    elif op.startswith('buffer_store') or op.startswith('global_store'): seq.append('S')
    elif op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('ds_read'): seq.append('r')
    elif op.startswith('ds_write'): seq.append('w')
    elif op.startswith('s_cbranch'): seq.append('|')
    elif op.startswith('s_barrier'): seq.append('B')
s = re.sub(r'(\| )+', '| ', ' '.join(seq))
print(s)
