"""GPU idle time inside a train step from a rocprofv3 --kernel-trace CSV (start/end timestamps of every kernel):
   rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
   python tools/gpu_idle.py out/t_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# take the last 60 % of the trace (steady-state steps)
t0 = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, n in ev[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = ev[-1][1] - ev[0][0]
print(f"kernels {len(ev)}, wall {wall/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {100*(wall-busy)/wall:.1f} %")
gaps.sort(reverse=True)
import collections
by = collections.Counter()
for g, n in gaps: by[n[:50]] += g
print("largest idle before:", [(round(g/1e3,1), n[:40]) for g, n in gaps[:8]])
print("idle by following kernel (us):", [(k, round(v/1e3)) for k, v in by.most_common(8)])
print("gap histogram (us): <2:%d 2-5:%d 5-10:%d 10-50:%d >50:%d" % (sum(g<2e3 for g,_ in gaps), sum(2e3<=g<5e3 for g,_ in gaps), sum(5e3<=g<1e4 for g,_ in gaps), sum(1e4<=g<5e4 for g,_ in gaps), sum(g>=5e4 for g,_ in gaps)))
