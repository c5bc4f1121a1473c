"""HBM bytes per train step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), per kernel family.
usage: hbm_traffic.py <fetch_dir> <write_dir> <steps_in_run> [workload label]
Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in kilobytes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so the fetch side is doubled; WRITE_SIZE is
taken as reported (uncalibrated, see the guide)."""
import csv, glob, json, os, re, sys
fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])


def load(d, counter):
    tot = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
    return tot


def family(name):
    if "gram_kernel" in name:
        return "bn"      # BatchNorm bookkeeping (statistics / backward sums of bn3 from conv3's input), not a layer
    if "igemm" in name or "wgrad" in name or "sconv" in name:
        return "conv"
    if "window_attn" in name:
        return "attention"
    if "layernorm" in name:
        return "layernorm"
    if name.startswith("bn_") or "bn_" in name[:40]:
        return "bn"
    return "other"


fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
fam = {}
for k in set(fe) | set(wr):
    f = family(re.sub(r"^void ", "", k))
    o = fam.setdefault(f, [0.0, 0.0])
    o[0] += fe.get(k, 0.0) * 1024 * 2      # KB -> B, x2 gfx950 correction
    o[1] += wr.get(k, 0.0) * 1024
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/collect_profiles.sh); FETCH_SIZE x2 (gfx950), KB units",
       "workload": sys.argv[4] if len(sys.argv) > 4 else "resnet50 bs256 bf16 224x224 train step", "steps_in_run": steps}
for f, (a, b) in fam.items():
    out[f + "_fetch_bytes_per_step"] = a / steps
    out[f + "_write_bytes_per_step"] = b / steps
out["conv_family_bytes_per_step"] = sum(fam.get("conv", [0, 0])) / steps
out["all_kernels_bytes_per_step"] = sum(a + b for a, b in fam.values()) / steps
# fingerprint of the kernel sources the counters were collected with (bench.py flags the file as stale when they change)
import hashlib
h = hashlib.sha256()
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pets-face-recognition_amd", "csrc")
for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
    h.update(open(f, "rb").read())
out["csrc_sha256"] = h.hexdigest()
print(json.dumps(out, indent=1))
