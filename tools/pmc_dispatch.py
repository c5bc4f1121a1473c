"""Per (kernel, grid size) averages of rocprofv3 --pmc counters WITH the dispatch duration (End - Start of the same csv rows), skipping the
first `skip` dispatches of each group (warm-up).   python tools/pmc_dispatch.py <dir> [name filter] [skip]
Derived columns when the counters are there: clock GHz = GRBM_GUI_ACTIVE / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x
busy CUs x GRBM_GUI_ACTIVE) is left to the reader (busy CUs = min(grid, 256))."""
import sys, csv, glob, os, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ''; skip = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = collections.defaultdict(dict)   # (kernel, grid, dispatch) -> {counter: value, '_dur': ns}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt and filt not in k:
            continue
        key = (k[:60], int(r.get("Grid_Size", 0) or 0), int(r["Dispatch_Id"]))
        rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            rows[key]['_dur'] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
groups = collections.defaultdict(list)
for (k, g, disp), v in sorted(rows.items()):
    groups[(k, g)].append(v)
for (k, g), lst in sorted(groups.items()):
    lst = lst[skip:] if len(lst) > skip else lst
    n = len(lst)
    avg = collections.defaultdict(float)
    for v in lst:
        for c, x in v.items():
            avg[c] += x / n
    dur = avg.pop('_dur', 0.0)
    extra = ''
    if dur and 'GRBM_GUI_ACTIVE' in avg:
        extra = f" clock {avg['GRBM_GUI_ACTIVE'] / dur:.3f} GHz"
    print(f"{k} grid {g} n {n} dur {dur / 1e3:.1f} us{extra} " + ' '.join(f"{c}={x:.5g}" for c, x in sorted(avg.items())))
