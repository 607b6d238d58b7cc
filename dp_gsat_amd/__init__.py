"""dp_gsat_amd -- MI355X (gfx950) implementation of GSAT's edge-attention + stochastic mask + masked
message-passing hot path behind the reference's PyG-style operator surface.

Public names follow the reference: ``get_model, GIN, PNA, GINConv, GINEConv, PNAConvSimple, MLP,
ExtractorMLP, GSAT, Criterion, get_preds, reorder_like, process_data``.
Everything computes through ``libgsat_hip.so`` (C ABI: include/gsat_hip.h); there is no CPU fallback.
"""
from .get_model import MLP, BatchSequential, Criterion, InstanceNorm, get_model, get_preds
from .conv_layers import GINConv, GINEConv, LEConv, PNAConvSimple
from .gin import GIN
from .pna import PNA
from .spmotif_gnn import SPMotifNet
from .gsat import (GSAT, ExtractorMLP, concrete_sample, get_r, gumbel_sigmoid, info_loss,
                   lift_node_att_to_edge_att, symmetrise_edge_att)
from .collate import PackedDataset, line_graph, line_graph_undirected
from .dual_gsat import DualGSAT, f1_sparsity_loss
from .graph_index import BatchIndex, clear_cache, get_index, set_strict, set_sync_free
from .utils import process_data, reorder_like, set_seed

__all__ = ["MLP", "BatchSequential", "Criterion", "InstanceNorm", "get_model", "get_preds", "GINConv", "GINEConv",
           "PNAConvSimple", "GIN", "PNA", "GSAT", "ExtractorMLP", "concrete_sample", "get_r", "gumbel_sigmoid",
           "info_loss", "lift_node_att_to_edge_att", "symmetrise_edge_att", "BatchIndex", "get_index", "clear_cache", "set_sync_free", "set_strict",
           "process_data", "reorder_like", "set_seed", "DualGSAT", "f1_sparsity_loss", "LEConv", "SPMotifNet", "PackedDataset", "line_graph", "line_graph_undirected"]
