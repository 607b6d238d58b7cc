"""Integer edge bookkeeping of the GSAT hot path, restated in numpy (ORACLE, test-only).

All results here are compared BIT-EXACTLY with the HIP kernels.
Indices are int64 at this level (the API type of the reference); the kernels
use int32 internally and tests compare after widening.
"""
from __future__ import annotations

import numpy as np


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def degree(index, num_nodes: int) -> np.ndarray:
    """[3P] torch_geometric.utils.degree: unweighted occurrence count.
    Call sites: src/models/conv_layers.py:183, src/utils/get_data_loaders.py:100."""
    return np.bincount(_np(index).astype(np.int64), minlength=num_nodes).astype(np.int64)


def deg_histogram(edge_index, num_nodes: int, minlength: int = 10) -> np.ndarray:
    """In-degree histogram handed to PNA as ``model_config['deg']``
    (src/utils/get_data_loaders.py:99-101: degree of edge_index[1], bincount minlength=10)."""
    d = degree(_np(edge_index)[1], num_nodes)
    return np.bincount(d, minlength=minlength).astype(np.int64)


def csr_by(rows, num_nodes: int):
    """Group edge slots by ``rows`` (stable): returns (rowptr[N+1], perm[E]) with
    perm = edge ids sorted by (rows, edge id).  This is the deterministic order the
    aggregation kernels sum in (MessagePassing.propagate aggregates at edge_index[1],
    src/models/conv_layers.py:21; order of summation is unspecified in the reference)."""
    rows = _np(rows).astype(np.int64)
    perm = np.argsort(rows, kind="stable").astype(np.int64)
    counts = np.bincount(rows, minlength=num_nodes)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, perm


def edge_keys(edge_index, num_nodes: int, transposed: bool = False) -> np.ndarray:
    ei = _np(edge_index).astype(np.int64)
    a, b = (ei[1], ei[0]) if transposed else (ei[0], ei[1])
    return a * np.int64(num_nodes) + b


def is_undirected(edge_index, num_nodes: int) -> bool:
    """[3P] torch_geometric.utils.is_undirected (example/gsat.py:80, src/run_gsat.py:232,242):
    True iff the sorted edge list equals the sorted transposed edge list (as multisets)."""
    k = np.sort(edge_keys(edge_index, num_nodes))
    kt = np.sort(edge_keys(edge_index, num_nodes, transposed=True))
    return bool(np.array_equal(k, kt))


def reverse_edge_perm(edge_index, num_nodes: int) -> np.ndarray:
    """rev[k] = position of the edge (dst_k, src_k).

    Restates transpose(coalesced=False) + reorder_like (example/gsat.py:81-82,
    src/utils/utils.py:19-25).  Duplicate edges make the reference's pairing
    ambiguous (unstable sorts); the build's documented rule is
    *stable* sorting on both sides: the i-th smallest (key, edge id) pairs with
    the i-th smallest (transposed key, edge id).  Raises ValueError exactly
    where reorder_like does (src/utils/utils.py:23-24)."""
    k = edge_keys(edge_index, num_nodes)
    kt = edge_keys(edge_index, num_nodes, transposed=True)
    p = np.argsort(k, kind="stable")
    q = np.argsort(kt, kind="stable")
    if not np.array_equal(k[p], kt[q]):
        raise ValueError("Edges in from_edge_index and to_edge_index are different, impossible to match both.")
    rev = np.empty(k.shape[0], dtype=np.int64)
    rev[q] = p
    return rev


def sort_edge_index(edge_index, num_nodes: int):
    """[3P] torch_geometric.utils.sort_edge_index: permutation sorting by row*N+col."""
    return np.argsort(edge_keys(edge_index, num_nodes), kind="stable")


def reorder_like(from_edge_index, to_edge_index, values):
    """src/utils/utils.py:19-25, line by line (numpy)."""
    f = _np(from_edge_index).astype(np.int64)
    t = _np(to_edge_index).astype(np.int64)
    v = _np(values)
    n = int(max(f.max(), t.max())) + 1 if f.size else 1
    order = sort_edge_index(f, n)                                  # :20
    f_sorted, v_sorted = f[:, order], v[order]
    ranking_score = t[0] * (t.max() + 1) + t[1]                    # :21
    ranking = np.argsort(np.argsort(ranking_score, kind="stable"), kind="stable")   # :22
    if not np.array_equal(f_sorted[:, ranking], t):                # :23
        raise ValueError("Edges in from_edge_index and to_edge_index are different, impossible to match both.")
    return v_sorted[ranking]                                       # :25


def graph_ptr(batch, num_graphs: int | None = None) -> np.ndarray:
    """Segment offsets of the non-decreasing PyG ``batch`` vector ([3P] Batch.from_data_list);
    number of segments = batch.max()+1 as in InstanceNorm / global pools."""
    b = _np(batch).astype(np.int64)
    g = int(b.max()) + 1 if num_graphs is None else num_graphs
    ptr = np.zeros(g + 1, dtype=np.int64)
    np.cumsum(np.bincount(b, minlength=g), out=ptr[1:])
    return ptr


def shard_graphs_lpt(edges_per_graph, world_size: int):
    """Edge-balanced whole-graph partition: greedy longest-processing-time,
    ties broken by lower graph id then lower rank; each rank's list is kept in ascending
    graph order so the local batch stays dataset-ordered."""
    e = _np(edges_per_graph).astype(np.int64)
    order = sorted(range(len(e)), key=lambda g: (-int(e[g]), g))
    loads = [0] * world_size
    out = [[] for _ in range(world_size)]
    for g in order:
        r = min(range(world_size), key=lambda i: (loads[i], i))
        out[r].append(g)
        loads[r] += int(e[g])
    return [sorted(x) for x in out]


def line_graph_by_source(edge_index):
    """Dual edges of the fork, restating src/datasets/mutag_dual.py:345-377: dual node = directed primal edge (by index);
    primal edges are grouped by their FIRST endpoint in order of first appearance (dict insertion order), and every pair
    i < j of a group contributes (e_i, e_j) then (e_j, e_i).  (The reference stores endpoint pairs and maps them to
    indices later; with duplicate-free edge lists that is the index pairing used here.)"""
    ei = _np(edge_index).astype(np.int64)
    groups = {}
    for idx in range(ei.shape[1]):
        groups.setdefault(int(ei[0, idx]), []).append(idx)
    out = []
    for group in groups.values():
        for i in range(len(group)):
            for j in range(i + 1, len(group)):
                out.append((group[i], group[j]))
                out.append((group[j], group[i]))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2).T


def line_graph_undirected(edge_index, batch, x=None, motif_start=20):
    """Dual graph of the fork's ba_2motifs dual dataset, restating src/datasets/ba_2motifs_dual.py:35-62 loop by loop on the
    dense adjacency matrix of every graph of a collated batch (the reference works on the dataset's [G, n, n] dense array):
      :42-50  scan (node1, node2) row-major; an un-numbered edge gets the next edge number in BOTH cells, its label is 1 iff
              node1 >= 20 and node2 >= 20 (motif nodes), its feature is x[node1] || x[node2];
      :52-57  for every row, all ordered pairs i != j of the edge numbers in that row become dual edges;
      :68     dense_to_sparse([3P]: nonzero in row-major order) lists them.
    Returns (dual_edge_index [2, E_d] with dual ids offset per graph, und_index [2, M] global endpoints, dual_batch [M],
    dual_x or None, dual_label [M])."""
    ei = _np(edge_index).astype(np.int64)
    b = _np(batch).astype(np.int64)
    xs = None if x is None else _np(x)
    ptr = graph_ptr(b)
    e_graph = b[ei[0]]
    dual_ei, und, dbatch, dx, dlabel = [], [], [], [], []
    off = 0
    for g in range(len(ptr) - 1):
        n0, n = int(ptr[g]), int(ptr[g + 1] - ptr[g])
        sel = e_graph == g
        graph = np.zeros((n, n), dtype=np.int64)
        graph[ei[0, sel] - n0, ei[1, sel] - n0] = 1                      # to_dense_adj
        graph -= 2                                                       # :40
        edge_num = 0
        for node1 in range(n):                                           # :44
            for node2 in range(n):
                if graph[node1][node2] == -1 and node1 != node2:         # :46
                    graph[node1][node2] = edge_num
                    graph[node2][node1] = edge_num
                    dlabel.append(1.0 if (node1 >= motif_start and node2 >= motif_start) else 0.0)   # :49-50
                    if xs is not None:
                        dx.append(np.concatenate((xs[n0 + node1], xs[n0 + node2]), axis=0))            # :51
                    und.append((n0 + node1, n0 + node2))
                    dbatch.append(g)
                    edge_num += 1
        dual_dense = np.zeros((edge_num, edge_num), dtype=np.int64)
        for row in graph:                                                # :55
            edges = row[row != -2]
            for i in edges:
                for j in edges:
                    if i != j:
                        dual_dense[int(i), int(j)] = 1
        r, c = np.nonzero(dual_dense)                                    # dense_to_sparse
        dual_ei.append(np.stack([r + off, c + off]))
        off += edge_num
    dei = np.concatenate(dual_ei, axis=1) if dual_ei else np.zeros((2, 0), dtype=np.int64)
    return (dei.astype(np.int64), np.asarray(und, dtype=np.int64).reshape(-1, 2).T, np.asarray(dbatch, dtype=np.int64),
            None if xs is None else np.stack(dx), np.asarray(dlabel, dtype=np.float32))
