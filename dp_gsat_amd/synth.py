"""Synthetic collated batches shaped like BASELINE.json's configs.

There is no network and no dataset on the GPU box, so bench.py and the parity tests draw PyG-style
batches ``(x, edge_index[2,E] int64, batch[N] int64, edge_attr, y)`` with the node / edge statistics of
the reference's datasets.  Generation is numpy on the host, seeded; tensors are moved to the device
by the caller.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


class Batch(SimpleNamespace):
    """Minimal stand-in for a collated PyG ``Batch`` (attribute access + ``.to(device)`` + ``.get``)."""

    def to(self, device):
        out = Batch()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if isinstance(v, torch.Tensor) else v)
        return out

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    @property
    def num_edges(self):
        return int(self.edge_index.shape[1])

    @property
    def num_nodes(self):
        return int(self.x.shape[0])


def _finish(src, dst, sizes, x, edge_attr, rng, num_class=2) -> Batch:
    G = len(sizes)
    batch = np.repeat(np.arange(G, dtype=np.int64), sizes)
    ei = torch.from_numpy(np.stack([np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)]))
    y = torch.from_numpy(rng.randint(0, 2, size=(G, 1)).astype(np.float32)) if num_class == 2 else \
        torch.from_numpy(rng.randint(0, num_class, size=(G,)).astype(np.int64))
    return Batch(x=x, edge_index=ei.contiguous(), batch=torch.from_numpy(batch), edge_attr=edge_attr, y=y, num_graphs=G)


def _both_directions(und_edges, off):
    s, d = [], []
    for (u, v) in und_edges:
        s += [u + off, v + off]
        d += [v + off, u + off]
    return s, d


def ba2motifs_batch(num_graphs=512, seed=0, x_dim=10) -> Batch:
    """C2: BA(20 nodes, m=1) + 5-node house or cycle + 1 link, both directions (src/datasets/ba_2motifs.py:29-31)."""
    rng = np.random.RandomState(seed)
    src, dst, sizes = [], [], []
    off = 0
    for _ in range(num_graphs):
        edges, deg = [(0, 1)], np.zeros(20)
        deg[0] = deg[1] = 1
        for v in range(2, 20):
            u = int(rng.choice(v, p=deg[:v] / deg[:v].sum()))
            edges.append((u, v)); deg[u] += 1; deg[v] += 1
        m = 20
        if rng.rand() < 0.5:      # house: square + roof
            edges += [(m, m + 1), (m + 1, m + 2), (m + 2, m + 3), (m + 3, m), (m + 4, m), (m + 4, m + 1)]
        else:                     # 5-cycle
            edges += [(m + i, m + (i + 1) % 5) for i in range(5)]
        edges.append((int(rng.randint(0, 20)), m))
        s, d = _both_directions(edges, off)
        src += s; dst += d; sizes.append(25); off += 25
    x = torch.from_numpy(rng.randn(off, x_dim).astype(np.float32))
    return _finish(src, dst, sizes, x, None, rng)


def molhiv_batch(num_graphs=2048, seed=0, categorical=True, x_dim=9) -> Batch:
    """C3: n ~ lognormal(mean ~25.5) clipped to [2,222]; random tree of max degree 4 plus ~12 % ring-closing
    edges, both directions (~55 directed edges per graph); x = 9 categorical atom columns."""
    from .encoders import ATOM_FEATURE_DIMS
    rng = np.random.RandomState(seed)
    sig = 0.45
    mu = np.log(25.5) - sig * sig / 2
    sizes = np.clip(np.round(rng.lognormal(mu, sig, size=num_graphs)), 2, 222).astype(np.int64)
    src, dst = [], []
    off = 0
    for n in sizes:
        n = int(n)
        deg = np.zeros(n, dtype=np.int64)
        edges = []
        for v in range(1, n):
            cand = np.flatnonzero(deg[:v] < 4)
            u = int(cand[rng.randint(len(cand))]) if len(cand) else int(rng.randint(v))
            edges.append((u, v)); deg[u] += 1; deg[v] += 1
        have = set(edges)
        want, tries = int(round(0.118 * n)), 0            # ring closures: ogbg-molhiv has ~27.5 bonds per 25.5 atoms (~55 directed edges)
        while want > 0 and tries < 40 * max(n // 8, 1):
            tries += 1
            a, b = sorted(int(t) for t in rng.randint(0, n, size=2))
            if a != b and (a, b) not in have and deg[a] < 4 and deg[b] < 4:
                edges.append((a, b)); have.add((a, b)); deg[a] += 1; deg[b] += 1
                want -= 1
        s, d = _both_directions(edges, off)
        src += s; dst += d; off += n
    if categorical:
        x = torch.from_numpy(np.stack([rng.randint(0, dmax, size=off) for dmax in ATOM_FEATURE_DIMS], axis=1).astype(np.int64))
    else:
        x = torch.from_numpy(rng.randn(off, x_dim).astype(np.float32))
    return _finish(src, dst, sizes, x, None, rng)


def spmotif_batch(num_graphs=4096, seed=0, x_dim=4) -> Batch:
    """C4: tree / ladder / wheel base + house / cycle / crane motif, ONE direction per edge
    (np.array(G.edges) of an nx.Graph, src/datasets/spmotif_utils/gen_spmotif.py:25,72,104), edge_attr = ones[E,1]."""
    rng = np.random.RandomState(seed)
    src, dst, sizes = [], [], []
    off = 0
    for _ in range(num_graphs):
        kind = rng.randint(3)
        if kind == 0:     # tree, ~32 nodes
            n = int(rng.randint(28, 37))
            edges = [(int(rng.randint(max(0, (v - 1) // 2), v)), v) for v in range(1, n)]
        elif kind == 1:   # ladder, ~24 nodes
            L = int(rng.randint(10, 14)); n = 2 * L
            edges = [(i, i + 1) for i in range(L - 1)] + [(L + i, L + i + 1) for i in range(L - 1)] + [(i, L + i) for i in range(L)]
        else:             # wheel, ~22 nodes
            n = int(rng.randint(18, 26))
            edges = [(0, i) for i in range(1, n)] + [(i, i + 1) for i in range(1, n - 1)] + [(1, n - 1)]
        m = n
        mk = rng.randint(3)
        if mk == 0:
            motif = [(m, m + 1), (m + 1, m + 2), (m + 2, m + 3), (m + 3, m), (m + 4, m), (m + 4, m + 1)]; k = 5
        elif mk == 1:
            motif = [(m + i, m + (i + 1) % 6) for i in range(6)]; k = 6
        else:             # crane
            motif = [(m, m + 1), (m, m + 2), (m, m + 3), (m + 1, m + 4), (m + 2, m + 4)]; k = 5
        edges = edges + motif + [(int(rng.randint(0, n)), m)]
        for (u, v) in edges:
            src.append(u + off); dst.append(v + off)
        sizes.append(n + k); off += n + k
    x = torch.from_numpy(rng.rand(off, x_dim).astype(np.float32))
    edge_attr = torch.ones(len(src), 1, dtype=torch.float32)
    return _finish(src, dst, sizes, x, edge_attr, rng, num_class=3)


def powerlaw_batch(num_nodes=1_250_000, num_edges=12_500_000, num_graphs=128, seed=0, x_dim=16, alpha=2.1,
                   max_deg=100_000) -> Batch:
    """C5 (per-GPU share by default): Chung-Lu graphs with Zipf(alpha) expected degrees clipped to [1, max_deg],
    cut into ``num_graphs`` equal node ranges; endpoints are drawn inside a graph, both directions kept."""
    rng = np.random.RandomState(seed)
    per = num_nodes // num_graphs
    sizes = np.full(num_graphs, per, dtype=np.int64)
    sizes[-1] += num_nodes - per * num_graphs
    w = np.clip(rng.zipf(alpha, size=num_nodes).astype(np.float64), 1, max_deg)
    und = num_edges // 2
    e_per = np.full(num_graphs, und // num_graphs, dtype=np.int64)
    e_per[-1] += und - e_per.sum()
    srcs, dsts = [], []
    off = 0
    for g in range(num_graphs):
        n = int(sizes[g])
        p = w[off:off + n] / w[off:off + n].sum()
        cdf = np.cumsum(p)
        need, got_a, got_b = int(e_per[g]), [], []
        while need > 0:                       # self pairs (frequent on hubs) are redrawn, so the edge total is the configured one
            a = np.minimum(np.searchsorted(cdf, rng.rand(need)), n - 1)
            b = np.minimum(np.searchsorted(cdf, rng.rand(need)), n - 1)
            keep = a != b
            got_a.append(a[keep]); got_b.append(b[keep])
            need -= int(keep.sum())
        a, b = np.concatenate(got_a) + off, np.concatenate(got_b) + off
        srcs += [a, b]; dsts += [b, a]
        off += n
    src = np.concatenate(srcs).astype(np.int64)
    dst = np.concatenate(dsts).astype(np.int64)
    x = torch.from_numpy(rng.randn(num_nodes, x_dim).astype(np.float32))
    G = num_graphs
    ei = torch.from_numpy(np.stack([src, dst]))
    return Batch(x=x, edge_index=ei.contiguous(), batch=torch.from_numpy(np.repeat(np.arange(G, dtype=np.int64), sizes)),
                 edge_attr=None, y=torch.from_numpy(rng.randint(0, 2, size=(G, 1)).astype(np.float32)), num_graphs=G)


def mutag_batch(npz_path: str, num_graphs: Optional[int] = None) -> Batch:
    """C1: real MUTAG topology + one-hot(14) node labels from the committed fixture (tests/golden/make_golden.py)."""
    z = np.load(npz_path)
    ei, batch, labels, y = z["edge_index"], z["batch"], z["node_label"], z["y"]
    G = int(batch.max()) + 1
    if num_graphs is not None and num_graphs < G:
        keep_n = batch < num_graphs
        n_keep = int(keep_n.sum())
        keep_e = ei[0] < n_keep
        ei, batch, labels, y, G = ei[:, keep_e], batch[:n_keep], labels[:n_keep], y[:num_graphs], num_graphs
    x = torch.zeros(len(labels), 14, dtype=torch.float32)
    x[torch.arange(len(labels)), torch.from_numpy(labels.astype(np.int64))] = 1.0
    return Batch(x=x, edge_index=torch.from_numpy(ei.astype(np.int64)).contiguous(), batch=torch.from_numpy(batch.astype(np.int64)),
                 edge_attr=None, y=torch.from_numpy(y.astype(np.float32)).view(-1, 1), num_graphs=G)


def mutag_full_topology(npz_path: str):
    """The whole Mutagenicity file of the reference (data/mutag_dual/raw: 4337 graphs, 131 488 nodes, 266 894 directed edges)
    from the committed fixture tests/golden/mutag_full.npz: (edge_index int64 [2,E] in file order, batch int64 [N], kept mask)."""
    z = np.load(npz_path)
    s = np.cumsum(z["even_src_delta"].astype(np.int64))
    d = s + z["even_dst_minus_src"].astype(np.int64)
    E = 2 * len(s)
    ei = np.empty((2, E), dtype=np.int64)
    ei[0, 0::2], ei[1, 0::2] = s, d            # edge 2k
    ei[0, 1::2], ei[1, 1::2] = d, s            # edge 2k+1 is its reverse
    sizes = z["nodes_per_graph"].astype(np.int64)
    batch = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)
    return torch.from_numpy(ei), torch.from_numpy(batch), torch.from_numpy(z["kept_mask"].astype(np.int64))


def in_degree_histogram(b: Batch, minlength: int = 10) -> torch.Tensor:
    """``deg`` handed to PNA: bincount of in-degrees, minlength 10 (src/utils/get_data_loaders.py:99-101)."""
    d = torch.bincount(b.edge_index[1].cpu(), minlength=b.num_nodes)
    return torch.bincount(d, minlength=minlength)
