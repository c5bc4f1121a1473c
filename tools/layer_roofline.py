"""Per-geometry roofline of the conv launches of one train step: bound time = max(FLOPs / MFMA peak, algorithmic bytes /
HBM peak) against the measured time (bench.py --detail, launches serialised by the event tracer).
Usage: python tools/layer_roofline.py gpurun_out/r02_detail.json [out.md]"""
import json
import re
import sys

PEAK_F, PEAK_B = 2.5e15, 8.0e12
rows = json.load(open(sys.argv[1]))  # bench.py --detail output
out = []
tot = {"ms": 0.0, "bound": 0.0, "mfma": 0.0, "hbm": 0.0, "flops": 0.0}
for r in rows:
    if r.get("mfma_bound_ms") is not None:     # rows already carry their bounds (bench.py computes them from the launch arguments)
        ms, t_m, t_h = r["ms_per_step"], r["mfma_bound_ms"], r["hbm_bound_ms"]
        b = max(t_m, t_h)
        tot["ms"] += ms; tot["bound"] += b; tot["mfma"] += t_m; tot["hbm"] += t_h; tot["flops"] += r["tflops"] * 1e9 * ms
        out.append((ms, f"| {r['op']} | {r['launches_per_step']} | {ms:.3f} | {r['tflops']:.0f} | {t_m:.3f} | {t_h:.3f} | {'hbm' if t_h > t_m else 'mfma'} | {b / ms:.2f} |"))
        continue
    m = re.match(r"(fwd|wgrad) N(\d+) H(\d+) W(\d+) C(\d+) Co(\d+) R(\d+) s(\d+)(?: dil(\d+))? OH(\d+)", r["op"])
    if not m:
        continue
    kind, N, H, W, C, Co, R, s, dil, OH = m.group(1), *[int(g) if g is not None else 0 for g in m.groups()[1:]]
    n = r["launches_per_step"]
    ms = r["ms_per_step"]
    flops = r["tflops"] * 1e12 * ms * 1e-3          # all launches of the row
    OW = OH * W // H if H else OH
    xin, yout = N * H * W * C * 2, N * OH * OW * Co * 2
    wbytes = Co * R * R * C * 2
    byts = n * (xin + yout + wbytes)
    t_m, t_h = flops / PEAK_F * 1e3, byts / PEAK_B * 1e3
    b = max(t_m, t_h)
    tot["ms"] += ms; tot["bound"] += b; tot["mfma"] += t_m; tot["hbm"] += t_h; tot["flops"] += flops
    out.append((ms, f"| {r['op']} | {n} | {ms:.3f} | {r['tflops']:.0f} | {t_m:.3f} | {t_h:.3f} | {'hbm' if t_h > t_m else 'mfma'} | {b / ms:.2f} |"))
out.sort(key=lambda t: -t[0])
lines = ["| launch geometry (bs 256, bf16) | launches/step | measured ms | TFLOP/s | MFMA-bound ms (2.5 PF) | HBM-bound ms (8 TB/s, activations+weights once) | binding | bound / measured |",
         "|---|---|---|---|---|---|---|---|"] + [o[1] for o in out]
lines.append(f"| **all conv launches listed** | | **{tot['ms']:.2f}** | {tot['flops'] / tot['ms'] / 1e9:.0f} | {tot['mfma']:.2f} | {tot['hbm']:.2f} | | **{tot['bound'] / tot['ms']:.3f}** |")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
