import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from pets_face_recognition_amd.data_loading import DeviceAugmentation
z = np.load('tests/golden/augment.npz')
tag = 'small'
crop, size = int(z[f"{tag}_crop"]), int(z[f"{tag}_size"])
aug = DeviceAugmentation((crop, crop), (size, size))
y = aug.apply(torch.from_numpy(z[f"{tag}_x"]).cuda(), torch.from_numpy(z[f"{tag}_flags"]), torch.from_numpy(z[f"{tag}_angles"])).cpu()
got = (y * 255).round().to(torch.uint8).numpy().transpose(0, 2, 3, 1)
want = z[f"{tag}_out"]
for i in range(got.shape[0]):
    d = got[i] != want[i]
    print(i, z[f"{tag}_flags"][i], z[f"{tag}_angles"][i], int(d.sum()), [int(d[..., c].sum()) for c in range(3)])
    if d.any():
        ys, xs, cs = np.nonzero(d)
        for k in range(min(6, len(ys))):
            print("   ", ys[k], xs[k], cs[k], got[i, ys[k], xs[k], cs[k]], want[i, ys[k], xs[k], cs[k]])
        x = z[f"{tag}_x"][i]
        print("   band lo/hi", [(int(x[..., c].min()), int(x[..., c].max())) for c in range(3)])
