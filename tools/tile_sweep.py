"""Tile-choice sweep for the conv / GEMM launches of a train step.
   1. python bench.py --detail gpurun_out/detail.json ...      (lists the distinct launch geometries of a step)
   2. python tools/tile_sweep.py gpurun_out/detail.json       (re-runs itself once per forced tile: PFR_TUNING=igemm_tile=<id>)
Prints, per geometry, the cold time of every tile variant and of the built-in heuristic."""
import sys, os, re, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = {-1: "heur", 0: "128x128", 1: "64x128", 2: "128x64", 3: "64x64", 4: "256x256", 5: "256x128"}
WTILES = {-1: "heur", 0: "128x128", 1: "64x128", 2: "128x64", 3: "64x64"}   # (Cout tile) x (K-column tile)
WGRAD = len(sys.argv) > 2 and "--wgrad" in sys.argv[2:]


def shapes(path):
    out = []
    for r in json.load(open(path)):
        if WGRAD:
            m = re.match(r"wgrad N(\d+) H(\d+) W(\d+) C(\d+) Co(\d+) R(\d+) s(\d+) OH(\d+) pro0", r["op"])
            if m and r["ms_per_step"] > 0.05:
                g = [int(v) for v in m.groups()]
                out.append((g[0], g[1], g[2], g[3], g[4], g[5], g[6], 0, g[7], r["launches_per_step"]))
            continue
        m = re.match(r"fwd N(\d+) H(\d+) W(\d+) C(\d+) Co(\d+) R(\d+) s(\d+) dil(\d+) OH(\d+) pro0", r["op"])
        if m and r["ms_per_step"] > 0.05:
            out.append(tuple(int(v) for v in m.groups()) + (r["launches_per_step"],))
    return out


def worker(path):
    import torch
    sys.path.insert(0, ROOT)
    from pets_face_recognition_amd._hip import ops
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    res = {}
    for (N, H, W, C, Co, R, s, dil, OH, n) in shapes(path):
        if C < 8 or (H == 1 and N <= 4096) or R not in (1, 3, 7):   # (patch-merging convs R = 2 / 4 are skipped)
            continue
        x = torch.randn(N, H, W, C, device="cuda").bfloat16()
        w = (torch.randn(Co, R, R, C, device="cuda") / (C * R * R) ** 0.5).bfloat16()
        pad = {1: 0, 3: 1, 7: 3}[R]
        t = 0.0
        if WGRAD:
            dy = torch.randn(N, OH, OH, Co, device="cuda").bfloat16()
            ws = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")
            out = ops.conv2d_wgrad(x, dy, R, R, s, pad, workspace=ws)
            run = lambda: ops.conv2d_wgrad(x, dy, R, R, s, pad, out=out, workspace=ws)
        else:
            kw = dict(stride=s, pad=pad, idil_log2=dil, out_hw=(OH, OH), stats=(dil == 0 and H > 1))   # (H = 1: the Linear layers of Swin, no BatchNorm statistics)
            y, part = ops.conv2d_fwd(x, w, **kw)
            run = lambda: ops.conv2d_fwd(x, w, out=y, stats_buf=part, **kw)
        for _ in range(4):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record()
            torch.cuda.synchronize(); t += a.elapsed_time(b) / 4
        res[f"H{H} C{C} Co{Co} R{R} s{s} dil{dil} x{n}"] = t * 1e3
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if "--worker" in sys.argv[2:]:
        worker(sys.argv[1])
        sys.exit(0)
    table = {}
    tiles = WTILES if WGRAD else TILES
    for tid, name in tiles.items():
        env = dict(os.environ)
        if tid >= 0:
            env["PFR_TUNING"] = ("wgrad_tile=" if WGRAD else "igemm_tile=") + str(tid)
        o = subprocess.run([sys.executable, __file__, sys.argv[1], "--worker"] + (["--wgrad"] if WGRAD else []), env=env,
                           capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith("RESULT ")]
        if line:
            for k, v in json.loads(line[0][7:]).items():
                table.setdefault(k, {})[name] = v
    tot_h = tot_b = 0.0
    for k, row in table.items():
        n = int(k.split("x")[-1])
        best = min((v, t) for t, v in row.items() if t != "heur")
        tot_h += row["heur"] * n; tot_b += best[0] * n
        print(f"{k:34s} " + " ".join(f"{t}:{row.get(t, float('nan')):7.1f}" for t in tiles.values()) + f"   best {best[1]} ({row['heur'] / best[0]:.2f}x)")
    print(f"per step: heuristic {tot_h / 1e3:.3f} ms, best-per-layer {tot_b / 1e3:.3f} ms")
