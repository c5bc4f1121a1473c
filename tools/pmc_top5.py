"""SQ counters of the N longest kernels of a workload: "where the matrix pipe waits", as a committed number every round.
usage: pmc_top5.py <dir of the SQ --pmc pass (with --kernel-trace)> [N = 5]
Per kernel (all its dispatches of the run): dispatches, total time (End - Start of the counter rows), and per dispatch:
  matrix-pipe busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x dispatch cycles at the clock GRBM_GUI_ACTIVE gives)   [0 for kernels without MFMA]
  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES  = share of a wave's life spent issue-stalled (MFMA dependency / busy pipe / LDS issue)
  SQ_WAIT_ANY / SQ_WAVE_CYCLES       = share parked at s_waitcnt / s_barrier
  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (when both are in the pass)"""
import sys, csv, glob, os, collections
d = sys.argv[1]; topn = int(sys.argv[2]) if len(sys.argv) > 2 else 5
disp = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], int(r["Dispatch_Id"]))
        disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("Start_Timestamp"):
            disp[key]["_dur"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ker = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for (k, _), v in disp.items():
    cnt[k] += 1
    for c, x in v.items():
        ker[k][c] += x
for k in sorted(ker, key=lambda k: -ker[k]["_dur"])[:topn]:
    v = ker[k]; n = cnt[k]; dur = v["_dur"]
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0           # summed over the 8 XCDs
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    line = f"{k[:90]}\n    dispatches {n}, total {dur / 1e6:.3f} ms, avg {dur / n / 1e3:.1f} us"
    if gui:
        line += f", clock (GRBM_GUI_ACTIVE / 8 / duration) {gui / dur:.2f} GHz"
        line += f", matrix pipe busy {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * gui):.3f}"
    if wc:
        line += f", issue-stalled {v.get('SQ_WAIT_INST_ANY', 0.0) / wc:.3f}, parked {v.get('SQ_WAIT_ANY', 0.0) / wc:.3f}, issuing {v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc:.3f} of the wave cycles"
    if v.get("SQ_LDS_IDX_ACTIVE"):
        line += f", LDS bank-conflict cycles / LDS active cycles {v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE']:.3f}"
    elif "SQ_LDS_BANK_CONFLICT" in v:
        line += f", LDS bank-conflict cycles per dispatch {v['SQ_LDS_BANK_CONFLICT'] / n:.3g}"
    print(line)
