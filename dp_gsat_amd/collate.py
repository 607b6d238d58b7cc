"""Device-resident dataset + batch assembly.

The reference collates on the host for every batch (PyG ``DataLoader`` -> ``Batch.from_data_list``,
src/utils/get_data_loaders.py:130-145) and then copies the batch to the device (src/run_gsat.py:654).  Here the whole
dataset is packed once into HBM (molhiv: 41 k graphs ~ 1 M nodes ~ 70 MB) and a batch is assembled by two small
kernels from a list of graph ids: no host work, no PCIe traffic per step.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from ._lib import call, ptr, stream
from .synth import Batch


class PackedDataset:
    def __init__(self, x_all, edge_index_local_all, node_ptr_all, edge_ptr_all, y_all, edge_attr_all=None, edge_label_all=None):
        self.x_all, self.edge_local_all = x_all.contiguous(), edge_index_local_all.contiguous()
        self.node_ptr_all, self.edge_ptr_all = node_ptr_all.contiguous(), edge_ptr_all.contiguous()
        self.y_all, self.edge_attr_all, self.edge_label_all = y_all, edge_attr_all, edge_label_all
        self.num_graphs = int(node_ptr_all.shape[0]) - 1
        self.node_counts = self.node_ptr_all[1:] - self.node_ptr_all[:-1]
        self.edge_counts = self.edge_ptr_all[1:] - self.edge_ptr_all[:-1]

    @classmethod
    def from_data_list(cls, graphs: Sequence, device) -> "PackedDataset":
        """graphs: objects with .x, .edge_index (local ids), .y and optionally .edge_attr / .edge_label (one-off, host side)."""
        n = torch.tensor([g.x.shape[0] for g in graphs], dtype=torch.int64)
        e = torch.tensor([g.edge_index.shape[1] for g in graphs], dtype=torch.int64)
        zero = torch.zeros(1, dtype=torch.int64)
        cat = lambda name: torch.cat([getattr(g, name) for g in graphs], dim=0).to(device) if getattr(graphs[0], name, None) is not None else None
        return cls(cat("x"), torch.cat([g.edge_index for g in graphs], dim=1).to(device), torch.cat([zero, n.cumsum(0)]).to(device),
                   torch.cat([zero, e.cumsum(0)]).to(device), cat("y"), cat("edge_attr"), cat("edge_label"))

    def collate(self, graph_ids: torch.Tensor, sizes: Optional[tuple] = None) -> Batch:
        """Batch of the graphs ``graph_ids`` (int64, on the device, any order).  ``sizes=(N, E)`` skips the one host sync
        that reads the batch's node / edge totals."""
        ids = graph_ids.to(self.x_all.device, torch.int64).contiguous()
        G = int(ids.shape[0])
        dev = ids.device
        zero = torch.zeros(1, dtype=torch.int64, device=dev)
        out_node_ptr = torch.cat([zero, self.node_counts[ids].cumsum(0)])
        out_edge_ptr = torch.cat([zero, self.edge_counts[ids].cumsum(0)])
        if sizes is None:
            N, E = (int(v) for v in torch.stack([out_node_ptr[-1], out_edge_ptr[-1]]).tolist())
        else:
            N, E = sizes
        batch = torch.empty(N, dtype=torch.int64, device=dev)
        node_src = torch.empty(N, dtype=torch.int64, device=dev)
        edge_index = torch.empty(2, E, dtype=torch.int64, device=dev)
        edge_src = torch.empty(E, dtype=torch.int64, device=dev)
        call("gsat_collate", ptr(ids), G, ptr(self.node_ptr_all), ptr(self.edge_ptr_all), ptr(self.edge_local_all),
             int(self.edge_local_all.shape[1]), ptr(out_node_ptr), ptr(out_edge_ptr), N, E, ptr(batch), ptr(node_src),
             ptr(edge_index), ptr(edge_src), stream())
        take = lambda t, idx: None if t is None else t.index_select(0, idx)
        return Batch(x=self.x_all.index_select(0, node_src), edge_index=edge_index, batch=batch, y=take(self.y_all, ids),
                     edge_attr=take(self.edge_attr_all, edge_src), edge_label=take(self.edge_label_all, edge_src), num_graphs=G)


def line_graph(edge_index: torch.Tensor, num_nodes: int, batch: Optional[torch.Tensor] = None):
    """Dual graph of the fork on the device: returns (dual_edge_index int64[2, E_d], dual_batch int64[E] or None).
    Dual node k is primal directed edge k (so primal edge attention and dual node attention align, src/run_gsat.py:253);
    dual edges join primal edges leaving the same node (src/datasets/mutag_dual.py:345-377)."""
    from .graph_index import get_index
    ix = get_index(edge_index, num_nodes)
    dev = edge_index.device
    counts = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    call("gsat_line_graph_pair_counts", ptr(ix.rowptr_src), num_nodes, ptr(counts), stream())
    pair_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts.cumsum(0)])
    num_pairs = int(pair_ptr[-1].item())
    dual_ei = torch.empty(2, 2 * num_pairs, dtype=torch.int64, device=dev)
    call("gsat_line_graph", ptr(ix.rowptr_src), ptr(ix.eid_by_src), ptr(pair_ptr), num_nodes, num_pairs, ptr(dual_ei), stream())
    dual_batch = None if batch is None else batch[edge_index[0]]
    return dual_ei, dual_batch


def line_graph_undirected(edge_index: torch.Tensor, num_nodes: int, batch: Optional[torch.Tensor] = None,
                          x: Optional[torch.Tensor] = None, motif_start: Optional[int] = None):
    """Dual graph with one node per UNDIRECTED primal edge, on the device -- the rule of the fork's ba_2motifs dual dataset
    (src/datasets/ba_2motifs_dual.py:35-62): edges numbered by (smaller, larger) endpoint in row-major order, dual nodes
    adjacent when the primal edges share an endpoint, dual edge list in (i, j) row-major order.

    Returns a ``Batch`` with ``edge_index`` (dual, int64 [2, E_d]), ``und_index`` (int64 [2, M]: endpoints a < b of every dual
    node), ``und_of_edge`` (int64 [E]: dual node of every primal directed edge, -1 for self loops), and when the inputs are
    given: ``batch`` (graph of every dual node), ``x`` = [x[a] || x[b]] (:48) and ``node_label`` = 1 iff both endpoints have
    local id >= ``motif_start`` (:46-47, 20 for BA-2motifs).  Raises ValueError if some edge has no reverse."""
    from .graph_index import call_size
    if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
        raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
    ei = edge_index.contiguous()
    dev, E, N = ei.device, int(ei.shape[1]), int(num_nodes)
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
    keys = torch.empty(max(E, 1), dtype=torch.int64, device=dev)          # uint64 keys: same bytes
    rowptr, und_of_slot, und_of_edge, und_src, und_dst = i32(N + 1), i32(E), i32(E), i32(E), i32(E)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    wb = max(call_size("gsat_und_edges_workspace_bytes", E), 256)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    call("gsat_und_edges", ptr(ei), E, N, ptr(keys), ptr(rowptr), ptr(und_of_slot), ptr(und_of_edge), ptr(und_src), ptr(und_dst),
         ptr(status), ptr(ws), wb, stream())
    M, asym, bad = status[:3].tolist()
    if bad:
        raise ValueError("edge_index contains node ids outside [0, num_nodes)")
    if asym:
        raise ValueError("line_graph_undirected needs a symmetric edge set: %d directed edges have no reverse" % asym)
    counts = torch.zeros(M, dtype=torch.int64, device=dev)
    call("gsat_und_line_graph_counts", ptr(rowptr), ptr(und_of_slot), ptr(und_src), ptr(und_dst), M, ptr(counts), stream())
    dual_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts.cumsum(0)])
    total = int(dual_ptr[-1].item())
    dual_ei = torch.empty(2, total, dtype=torch.int64, device=dev)
    call("gsat_und_line_graph", ptr(rowptr), ptr(und_of_slot), ptr(und_src), ptr(und_dst), ptr(dual_ptr), M, total, ptr(dual_ei), stream())
    a, b = und_src[:M].long(), und_dst[:M].long()
    out = Batch(edge_index=dual_ei, und_index=torch.stack([a, b]), und_of_edge=und_of_edge[:E].long(), batch=None, x=None,
                node_label=None, edge_attr=None, num_dual_nodes=M)
    if batch is not None:
        out.batch = batch[a]
    if x is not None:
        out.x = torch.cat([x[a], x[b]], dim=1)
    if motif_start is not None:
        if batch is None:
            raise ValueError("motif labels need the batch vector (local node ids)")
        G = int(batch.max().item()) + 1 if batch.numel() else 0
        node_ptr = torch.zeros(G + 1, dtype=torch.int64, device=dev)
        node_ptr[1:] = torch.bincount(batch, minlength=G).cumsum(0)
        la, lb = a - node_ptr[batch[a]], b - node_ptr[batch[b]]
        out.node_label = ((la >= motif_start) & (lb >= motif_start)).float()
    return out
