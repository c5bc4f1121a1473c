"""`PairGenerator` — the verification-pair sampler the evaluation consumes (/root/reference/data_loading/pairs.py:10-108).

Seeded-equivalent to the reference: the same `(dataset.uid_to_indices, gen_number, gen_ratio, random_seed, usr_list)`
yields the same `pairs` list and the same `correction` table (pinned by tests/golden/pairs.npz, which
oracle/make_golden.py writes by running the reference's class).  What has to coincide for that:

  * one `np.random.RandomState(random_seed)` consumed in the reference's order: per identity (dict order of
    `uid_to_indices`) one `choice(len(candidates), n, replace=False)` over the genuine candidates, then — in a second
    sweep — one over the impostor candidates (pairs.py:50-74);
  * candidate enumeration order: genuine = ordered pairs (ii, jj), ii != jj, row-major over the identity's index list;
    impostor = (ii, jj) for ii in own indices, for jj in the SET `all_indices - own` (set iteration order, reproduced
    here by building the set the same way);
  * per-identity quota `min(round(part / max * wanted), part)` with Python's banker's `round` (pairs.py:53,69);
  * `correction[i]` = rank of i among the sorted indices of the selected users (pairs.py:76-86 computes exactly that by
    accumulating the gaps): the evaluator sorts embeddings by dataset index (engine/controller.py:51-56), so
    `corrected_indices` address rows of the sorted embedding matrix.

The image-pair `__getitem__` (pairs.py:20-25) and the pickle cache are kept; file-system scanning is the dataset's job.
"""
import pickle
from pathlib import Path

import numpy as np


class PairGenerator:
    def __init__(self, dataset, gen_number=None, gen_ratio=1, path=None, random_seed=None, usr_list=None):
        self.dataset = dataset
        if path is None or not Path(path).exists():
            self.generate_pairs(gen_number, gen_ratio, path, random_seed, usr_list)
        else:
            with open(path, 'rb') as f:
                self.pairs, self.correction = pickle.load(f)

    def __getitem__(self, item):
        a, b, lab = self.pairs[item]
        return {'x1': self.dataset[a]['x'], 'x2': self.dataset[b]['x'], 'label': int(lab)}

    def __len__(self):
        return len(self.pairs)

    def generate_pairs(self, gen_number, gen_ratio, path, random_seed, usr_list):
        rand = np.random.RandomState(random_seed)
        u2i = self.dataset.uid_to_indices
        total = len(self.dataset)
        users = set(usr_list) if usr_list is not None else set(u2i)
        sel = [(u, idx) for u, idx in u2i.items() if u in users]          # dict order, as the reference iterates it

        max_gen = sum(len(idx) * len(idx) - len(idx) for _, idx in sel)
        max_imp = sum(total * len(idx) - min(total, len(idx)) for _, idx in sel)
        if gen_number is not None:
            assert gen_number <= max_gen, f'{gen_number} greater than {max_gen}'
        else:
            gen_number = max_gen
        imp_number = int(gen_number * gen_ratio)
        assert imp_number <= max_imp, f'{imp_number} greater than {max_imp}'

        genuine = []
        for _, idx in sel:
            if len(idx) <= 1:
                continue
            part = len(idx) * len(idx) - len(idx)
            n = min(round(part / max_gen * gen_number), part)
            cand = [(a, b) for a in idx for b in idx if a != b]
            genuine.extend(cand[k] for k in rand.choice(len(cand), n, replace=False))

        impostor = []
        all_indices = {j for _, idx in sel for j in idx}
        for _, idx in sel:
            part = total * len(idx) - min(total, len(idx))
            n = min(round(part * imp_number / max_imp), part)
            others = all_indices - set(idx)
            cand = [(a, b) for a in idx for b in others]
            impostor.extend(cand[k] for k in rand.choice(len(cand), n, replace=False))

        self.correction = {i: r for r, i in enumerate(sorted(all_indices))}
        self.pairs = [(a, b, 1) for a, b in genuine] + [(a, b, 0) for a, b in impostor]
        if path is not None:
            with open(path, 'wb') as f:
                pickle.dump([self.pairs, self.correction], f)

    @property
    def labels(self):
        return np.array([int(lab) for _, _, lab in self.pairs])

    @property
    def indices(self):
        return [(a, b) for a, b, _ in self.pairs]

    @property
    def corrected_indices(self):
        return [(self.correction[a], self.correction[b]) for a, b, _ in self.pairs]
