"""`PairGenerator` — the verification-pair interface the evaluation consumes
(/root/reference/data_loading/pairs.py:10-108): `labels` (1 genuine / 0 impostor), `indices` (dataset indices) and
`corrected_indices` (= positions inside the SORTED validation-index set, because the controller sorts embeddings by
dataset index before scoring, engine/controller.py:51-56).  Seeded sampling of genuine / impostor pairs over the
validation identities; the file-system scanning parts of the reference are out of scope."""
import random


class PairGenerator:
    def __init__(self, dataset, n_genuine, impostor_ratio=1, _unused=None, seed=0, users=None):
        rng = random.Random(seed)
        labels = dataset.get_labels()
        users = set(users) if users is not None else set(labels)
        by_user = {}
        for idx, u in enumerate(labels):
            if u in users:
                by_user.setdefault(u, []).append(idx)
        multi = [u for u, v in by_user.items() if len(v) > 1]
        all_users = list(by_user)
        assert multi and len(all_users) > 1
        pairs, plabels = [], []
        seen = set()
        tries = 0
        while sum(plabels) < n_genuine and tries < 50 * n_genuine:
            tries += 1
            u = rng.choice(multi)
            a, b = rng.sample(by_user[u], 2)
            key = (min(a, b), max(a, b))
            if key not in seen:
                seen.add(key)
                pairs.append(key)
                plabels.append(1)
        n_imp = int(sum(plabels) * impostor_ratio)
        tries = 0
        while len(pairs) - sum(plabels) < n_imp and tries < 50 * n_imp:
            tries += 1
            u, v = rng.sample(all_users, 2)
            key = (rng.choice(by_user[u]), rng.choice(by_user[v]))
            key = (min(key), max(key))
            if key not in seen:
                seen.add(key)
                pairs.append(key)
                plabels.append(0)
        self._pairs, self._labels = pairs, plabels
        val_sorted = sorted(i for v in by_user.values() for i in v)
        self._rank = {idx: r for r, idx in enumerate(val_sorted)}

    def __len__(self):
        return len(self._pairs)

    @property
    def labels(self):
        return list(self._labels)

    @property
    def indices(self):
        return list(self._pairs)

    @property
    def corrected_indices(self):
        return [(self._rank[a], self._rank[b]) for a, b in self._pairs]
