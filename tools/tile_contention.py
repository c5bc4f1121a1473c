"""What slows a 256x256 tile down when more CUs are busy?     One configuration per invocation, so that a
rocprofv3 --pmc pass sees only that configuration's dispatches:

  python tools/tile_contention.py fwd  <case> <N> [data=randn|zeros] [krot=K] [dbg=F] [reps=R] [secs=S]
  python tools/tile_contention.py wgrad <case> <N> [splits=S] ...

data=zeros   zero-filled operands (DVFS: the chip clocks higher when the MFMAs toggle nothing)
krot=K       pfr_set_tuning("igemm_krot", K): workgroup t starts its k-loop at k-step (t * K) % nk
dbg=16       (trace library, PFR_LIB_PATH=.../libpfr_hip_trace.so) the tiles run on XCDs 0-3 only
secs=S       instead of `reps` launches keep launching for S seconds (for an smi power / clock sample next to it)
Prints one line: case, N, tiles, us per launch, TF/s."""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib

FWD = {'c3x3_256_h14': (14, 256, 256, 3, 1), 'c1x1_1024_256_h14': (14, 1024, 256, 1, 0), 'c3x3_512_h7': (7, 512, 512, 3, 1),
       'c3x3_128_h28': (28, 128, 128, 3, 1), 'c1x1_256_1024_h14': (14, 256, 1024, 1, 0)}
kind, case = sys.argv[1], sys.argv[2]
NS = [int(v) for v in sys.argv[3].split(',')]
kw = dict(a.split('=') for a in sys.argv[4:])
DATAS = kw.get('data', 'randn').split(','); KROTS = [int(v) for v in kw.get('krot', '0').split(',')]
dbg = int(kw.get('dbg', 0)); reps = int(kw.get('reps', 30))
secs = float(kw.get('secs', 0)); SPLITS = [int(v) for v in kw.get('splits', '0').split(',')]
DMAS = [int(v, 0) for v in kw.get('dma', '0x101').split(',')]
TILES = [int(v) for v in kw.get('tile', '-1').split(',')]   # pfr_set_tuning("igemm_tile"): 4 = the 8-wave 256x256 tile whatever the tile count
if dbg:
    dll = ctypes.CDLL(os.environ['PFR_LIB_PATH'])
    dll.pfr_debug_igemm_flags(dbg)
H, C, Co, R, p = FWD[case]


def run(N, data, krot, splits, dma=0x101, tile=-1):
    lib.pfr_set_tuning(b"igemm_dma", dma)
    lib.pfr_set_tuning(b"igemm_krot", krot)
    lib.pfr_set_tuning(b"igemm_tile", tile)
    lib.pfr_set_tuning(b"wgrad_splits", splits)
    mk = (lambda *s: torch.zeros(*s, device='cuda')) if data == 'zeros' else (lambda *s: torch.randn(*s, device='cuda'))
    x = mk(N, H, H, C).bfloat16()
    if kind == 'fwd':
        w = (mk(Co, R, R, C) / (C * R * R) ** 0.5).bfloat16()
        y, part = ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True)
        go = lambda: ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True, out=y, stats_buf=part)
    else:
        dy = mk(N, H, H, Co).bfloat16()
        ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device='cuda')
        out = ops.conv2d_wgrad(x, dy, R, R, 1, p, workspace=ws)
        go = lambda: ops.conv2d_wgrad(x, dy, R, R, 1, p, out=out, workspace=ws)
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    if secs > 0:
        tw = time.time(); n = 0
        t0.record()
        while time.time() - tw < secs:
            for _ in range(200):
                go()
            n += 200
            torch.cuda.synchronize()
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / n
    else:
        t0.record()
        for _ in range(reps):
            go()
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
    M = N * H * H
    fl = 2.0 * M * Co * R * R * C
    tiles = (M + 255) // 256 * ((Co + 255) // 256)
    chk = int(y.view(torch.int16).to(torch.int64).sum().item()) if kind == 'fwd' else 0
    print(f'{kind:5s} {case:18s} N {N:4d} tiles256 {tiles:4d} data {data:5s} dma {dma:#05x} chk {chk:14d} tile {tile:2d} krot {krot:2d} dbg {dbg:2d} splits {splits:2d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s', flush=True)


for N in NS:
    for data in DATAS:
        for krot in KROTS:
            for sp in SPLITS:
                for dma in DMAS:
                    for tile in TILES:
                        run(N, data, krot, sp, dma, tile)
