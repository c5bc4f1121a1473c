"""Synthetic stand-in for the reference's folder-per-identity `RecDataset` (/root/reference/data_loading/dataset.py:
67-201): same item contract `{'x': float 3xHxW in [0,1], 'label': int64, 'index': int64}` (dataset.py:125), same
`get_users()/get_labels()` surface used by the configs, but images are generated (datasets need network downloads).
Each identity has a fixed random low-frequency pattern; photos are the pattern plus noise, so that embeddings can
actually separate identities."""
import torch
from torch.utils.data import Dataset


class SyntheticRecDataset(Dataset):
    def __init__(self, n_identities, photos_per_identity, image_size=224, seed=0, noise=0.15, raw_uint8=False, noise_bank=0):
        self.n_id, self.ppi, self.size, self.seed, self.noise = n_identities, photos_per_identity, image_size, seed, noise
        # noise_bank = K > 0: the per-photo noise comes from K pre-drawn frames (photo i uses frame i % K) instead of a fresh
        # 150 k-sample draw per item — a loader whose per-item cost is that of a cached, already decoded frame (throughput runs)
        self.noise_bank = None
        if noise_bank:
            g = torch.Generator().manual_seed(seed * 104729 + 17)
            self.noise_bank = noise * torch.randn(int(noise_bank), 3, image_size, image_size, generator=g)
        # raw_uint8: hand out the HWC uint8 frame the reference's dataset holds BEFORE its transform (dataset.py:100-121);
        # the augmentation then runs on the device for the whole batch (data_loading/augment.py)
        self.raw_uint8 = raw_uint8
        self.labels = torch.arange(n_identities).repeat_interleave(photos_per_identity)
        self.label_map = {u: u for u in range(n_identities)}
        # identity -> dataset indices, the attribute the reference's PairGenerator samples from (dataset.py:93-96)
        self.uid_to_indices = {u: list(range(u * photos_per_identity, (u + 1) * photos_per_identity)) for u in range(n_identities)}

    def __len__(self):
        return self.n_id * self.ppi

    def get_users(self):
        return list(range(self.n_id))

    def get_labels(self):
        return self.labels.tolist()

    def _pattern(self, ident):
        g = torch.Generator().manual_seed(self.seed * 1000003 + ident)
        low = torch.rand(3, 7, 7, generator=g)
        return torch.nn.functional.interpolate(low[None], size=(self.size, self.size), mode='bilinear', align_corners=False)[0]

    def __getitem__(self, i):
        ident = int(self.labels[i])
        if self.noise_bank is not None:
            x = (self._pattern(ident) + self.noise_bank[i % self.noise_bank.shape[0]]).clamp_(0, 1)
        else:
            g = torch.Generator().manual_seed(self.seed * 7919 + i)
            x = (self._pattern(ident) + self.noise * torch.randn(3, self.size, self.size, generator=g)).clamp_(0, 1)
        if self.raw_uint8:
            x = (x * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous()
        return {'x': x, 'label': torch.tensor(self.label_map[ident], dtype=torch.int64), 'index': torch.tensor(i, dtype=torch.int64)}


class RecSubset(Dataset):
    def __init__(self, dataset, indices):
        self.dataset, self.indices = dataset, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.dataset[self.indices[i]]
