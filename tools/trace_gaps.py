"""Idle time between kernels of a rocprofv3 --kernel-trace run: python tools/trace_gaps.py <dir> [skip_first_n_kernels]
Prints busy time (union of kernel intervals), the idle gaps between them, and the gap histogram, over the LAST 40 % of the trace
(the timed steps of a bench.py run)."""
import sys, csv, glob, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
rows.sort()
n = len(rows)
rows = rows[int(n * 0.6):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []; where = []; last = rows[0][2]
for s, e, k in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); where.append((s - cur_e, last, k)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last = k
busy += cur_e - cur_s
tot = t1 - t0
print(f"kernels {len(rows)}  span {tot / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / tot:.1f} %)  idle {sum(gaps) / 1e6:.2f} ms in {len(gaps)} gaps")
import collections
h = collections.Counter()
for g in gaps:
    h[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count):", dict(sorted(h.items())))
big = sorted(gaps)[-5:]
print("largest gaps (us):", [round(g / 1e3, 1) for g in big], " median gap (us):", round(sorted(gaps)[len(gaps) // 2] / 1e3, 2))
for g, a, b in sorted(where)[-8:]:
    print(f"  {g / 1e3:7.1f} us between {a}  ->  {b}")
import collections as _c
pair = _c.Counter()
for g, a, b in where:
    pair[(a[:36], b[:36])] += g
print("most idle time by (kernel before -> kernel after), us over the window:")
for (a, b), g in pair.most_common(8):
    print(f"  {g / 1e3:8.1f}  {a} -> {b}")
