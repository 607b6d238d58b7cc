"""Edge-bookkeeping helpers with the reference's names (src/utils/utils.py:19-33, 75-82)."""
from __future__ import annotations

import random

import numpy as np
import torch



def reorder_like(from_edge_index, to_edge_index, values):
    """values re-ordered from ``from_edge_index``'s edge order to ``to_edge_index``'s (src/utils/utils.py:19-25).
    Raises the reference's ValueError when the two edge sets differ; duplicate edges pair in stable order.
    Kept for API parity only: the GSAT step itself uses the cached ``BatchIndex.rev`` permutation."""
    if from_edge_index.shape != to_edge_index.shape:
        raise ValueError("Edges in from_edge_index and to_edge_index are different, impossible to match both.")
    if from_edge_index.shape[1] == 0:
        return values
    n = int(torch.maximum(from_edge_index.max(), to_edge_index.max()).item()) + 1
    key_f = from_edge_index[0] * n + from_edge_index[1]
    key_t = to_edge_index[0] * n + to_edge_index[1]
    pf = torch.sort(key_f, stable=True)[1]
    pt = torch.sort(key_t, stable=True)[1]
    if not torch.equal(key_f[pf], key_t[pt]):
        raise ValueError("Edges in from_edge_index and to_edge_index are different, impossible to match both.")
    match = torch.empty_like(pf)
    match[pt] = pf
    return values[match]


def process_data(data, use_edge_attr):
    """src/utils/utils.py:28-33."""
    if not use_edge_attr:
        data.edge_attr = None
    if data.get("edge_label", None) is None:
        data.edge_label = torch.zeros(data.edge_index.shape[1])
    return data


def set_seed(seed):
    """src/utils/utils.py:75-82."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
