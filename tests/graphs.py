"""Small seeded graph batches shared by the tests (inputs only; no reference code)."""
import numpy as np
import torch


def random_batch(seed, num_graphs, n_lo, n_hi, undirected=True, extra_edge_frac=0.15, isolated=True):
    """Batched random trees + a few ring-closing edges (molecule-like), optionally with isolated nodes,
    as a PyG-style (edge_index[2,E] int64, batch[N] int64)."""
    rng = np.random.RandomState(seed)
    src, dst, batch = [], [], []
    off = 0
    for g in range(num_graphs):
        n = int(rng.randint(n_lo, n_hi + 1))
        es = set()
        for v in range(1, n):
            if isolated and rng.rand() < 0.05:
                continue                      # leave v without a parent edge (may stay isolated)
            u = int(rng.randint(0, v))
            es.add((u, v))
        for _ in range(int(extra_edge_frac * n)):
            a, b = rng.randint(0, n, size=2)
            if a != b and (min(a, b), max(a, b)) not in es:
                es.add((int(min(a, b)), int(max(a, b))))
        for (u, v) in sorted(es):
            src.append(u + off); dst.append(v + off)
            if undirected:
                src.append(v + off); dst.append(u + off)
        batch += [g] * n
        off += n
    ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)
    return ei, torch.tensor(batch, dtype=torch.int64), off


def shuffle_edges(ei, seed):
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(ei.shape[1], generator=g)
    return ei[:, perm].contiguous()


def line_graph(ei, batch):
    """Dual graph with one node per DIRECTED primal edge (as mutag_dual: N_dual = E_primal); two dual nodes are joined
    (both directions) when their primal edges share an endpoint and are not the same undirected edge."""
    import itertools
    src, dst = ei[0].tolist(), ei[1].tolist()
    E = len(src)
    inc = {}
    for k in range(E):
        inc.setdefault(src[k], []).append(k)
        inc.setdefault(dst[k], []).append(k)
    es = set()
    for _, ks in inc.items():
        for a, b in itertools.combinations(sorted(set(ks)), 2):
            if {src[a], dst[a]} != {src[b], dst[b]}:
                es.add((a, b)); es.add((b, a))
    es = sorted(es)
    dei = torch.tensor([[a for a, _ in es], [b for _, b in es]], dtype=torch.int64).reshape(2, -1)
    dbatch = batch[ei[0]].clone()
    assert bool((dbatch[:-1] <= dbatch[1:]).all())
    return dei, dbatch
