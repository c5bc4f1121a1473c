"""Kernel timeline of ONE training step from a rocprofv3 --kernel-trace run of bench.py (side stream on):
   python tools/step_timeline.py <trace dir> <out.tsv> [marker kernel substring = s2d_input]
Writes every kernel of the second-to-last step (queue, start us, duration us, gap to the previous kernel of the same queue, name) and prints:
step span, busy time per queue, union busy / idle, the main queue's gaps by size class and by (kernel before -> kernel after), and what the
non-library launches (copyBuffer, fill, elementwise) sit between."""
import sys, csv, glob, os, collections
d, out = sys.argv[1], sys.argv[2]
mark = sys.argv[3] if len(sys.argv) > 3 else "s2d_input"
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
byq = collections.defaultdict(float); cntq = collections.Counter()
for r in step:
    byq[r["Queue_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cntq[r["Queue_Id"]] += 1
mainq = max(byq, key=byq.get)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels; busy per queue (us): " + ", ".join(f"{q}: {v:.0f} ({cntq[q]} launches)" for q, v in byq.items()))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cs, ce = iv[0]; busy = 0
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"union busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
last = {}; lines = []; gaps = []; prev = {}
for r in step:
    q = r["Queue_Id"]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - last[q]) / 1e3 if q in last else 0.0
    name = r["Kernel_Name"]
    short = name.split("(")[0][-60:] if not name.startswith("_Z") else name[:70]
    lines.append(f"{'M' if q == mainq else 'S'}\t{(s - t0) / 1e3:9.1f}\t{(e - s) / 1e3:7.1f}\t{g:7.1f}\t{r.get('Grid_Size', '')}\t{short}")
    if q == mainq and q in last:
        gaps.append((g, prev[q], short))
    last[q] = e; prev[q] = short
open(out, "w").write("queue\tstart_us\tdur_us\tgap_us\tgrid\tkernel\n" + "\n".join(lines) + "\n")
cls = collections.Counter(); tot = collections.Counter()
for g, a_, b_ in gaps:
    k = "<1" if g < 1 else "1-2" if g < 2 else "2-4" if g < 4 else "4-8" if g < 8 else "8-20" if g < 20 else ">20"
    cls[k] += 1; tot[k] += g
print("main-queue gaps by size (us): " + ", ".join(f"{k}: {cls[k]} gaps / {tot[k]:.0f} us" for k in ("<1", "1-2", "2-4", "4-8", "8-20", ">20")) +
      f"; all positive gaps {sum(g for g, _, _ in gaps if g > 0):.0f} us")
pair = collections.Counter(); pcnt = collections.Counter()
for g, a_, b_ in gaps:
    if g > 0:
        pair[(a_[:40], b_[:40])] += g; pcnt[(a_[:40], b_[:40])] += 1
print("main-queue idle by (kernel before -> kernel after):")
for (a_, b_), g in pair.most_common(25):
    print(f"  {g:8.1f} us in {pcnt[(a_, b_)]:3d}  {a_} -> {b_}")
