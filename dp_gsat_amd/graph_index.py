"""Per-batch integer bookkeeping, built once on the device and cached.

The reference re-derives all of this on every forward (a coalesce/sort in ``is_undirected``, a sort in
``sort_edge_index``, two ``argsort`` in ``reorder_like`` -- example/gsat.py:80-83,
src/utils/utils.py:19-25 -- and a scatter index per ``propagate`` call, src/models/conv_layers.py:21).
Here a collated batch gets one :class:`BatchIndex` holding int32 CSR views of ``edge_index`` by
destination and by source, the reverse-edge permutation and the per-graph segment pointers.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib
from ._lib import call, ptr, stream


# Sync-free mode: nothing in the step reads device memory back on the host (no `if is_undirected`, no hub-row count),
# so a whole step can be captured into a hipGraph (torch.cuda.graph) and replayed; decisions move into the kernels.
_SYNC_FREE = False
_VALIDATED: set = set()       # batches already validated in sync-free mode (BatchIndex._validate_once)



# Hub rows without a host sync.  Whether a batch has rows above GSAT_LONG_ROW_EDGES is a device-side fact (status words [1], [2]).
# The GIN ops read it back (one small sync per batch, as the reference's `is_undirected` does anyway); the PNA ops must not stall the
# launch queue of a 1 ms step for it, so every new index queues an asynchronous copy of its status words into a pinned slot, later
# indices harvest the copies that have landed, and `long_rows_nowait` answers from what is known: the exact flags once this batch's
# copy (or a read-back) has arrived, otherwise "hubs possible" iff an earlier batch of the process had any.
_HUBS_SEEN = [False]
_STATUS_RING = []          # [pinned int32[8], event, owner weakref | None]


def _harvest_status():
    bad = None
    for slot in _STATUS_RING:
        host, ev, owner = slot
        if owner is not None and ev.query():
            vals = host.tolist()
            ix = owner()
            if ix is not None and ix._long is None:
                ix._long_async = (vals[1] > 0, vals[2] > 0)
            if vals[1] > 0 or vals[2] > 0:
                _HUBS_SEEN[0] = True
            slot[2] = None
            if ix is None or not ix._checked:          # nobody read this batch's status back: report its errors now, late but not never
                if vals[0] != 0:
                    bad = "edge_index of an earlier batch contained node ids outside [0, num_nodes) (reported late: the PNA path never syncs)"
                elif vals[3] != 0:
                    bad = "`batch` of an earlier batch was not non-decreasing with ids in [0, num_graphs) (reported late: the PNA path never syncs)"
    if bad:
        raise ValueError(bad)


def _queue_status(index):
    import weakref
    if os.environ.get("GSAT_STATUS_ASYNC", "1") == "0":          # A/B switch: no status copies (hubs stay unknown on the PNA path)
        return
    _harvest_status()
    slot = next((s for s in _STATUS_RING if s[2] is None), None)
    if slot is None:
        if len(_STATUS_RING) >= 16:
            return
        slot = [torch.empty(8, dtype=torch.int32).pin_memory(), torch.cuda.Event(), None]
        _STATUS_RING.append(slot)
    slot[0].copy_(index._err, non_blocking=True)
    slot[1].record()
    slot[2] = weakref.ref(index)
    index._status_event = slot[1]


# Strict mode (debugging): the PNA ops normally learn about hub rows and bad ids through an asynchronous status copy, so a batch with
# node ids outside [0, num_nodes) or an unsorted `batch` vector raises its ValueError one batch LATE on that path (the kernels clamp ids:
# memory-safe, results of that batch meaningless).  set_strict(True) restores the reference's behaviour -- an immediate ValueError -- at
# the price of one device->host read per batch in front of the first PNA launch.
_STRICT = False


def set_strict(flag: bool) -> None:
    global _STRICT
    _STRICT = bool(flag)


def strict() -> bool:
    return _STRICT


def set_sync_free(flag: bool) -> None:
    global _SYNC_FREE
    _SYNC_FREE = bool(flag)


def sync_free() -> bool:
    return _SYNC_FREE


def _i32(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.int32, device=device)[: int(n)]


class BatchIndex:
    def __init__(self, edge_index: torch.Tensor, num_nodes: int):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        _lib.load()
        if not edge_index.is_cuda:
            raise _lib.GsatHipError("BatchIndex needs a ROCm edge_index (no CPU fallback)")
        self._key_tensor = edge_index          # keeps the keyed storage alive while cached
        self.edge_index = edge_index.contiguous()
        self.N = int(num_nodes)
        self.E = int(edge_index.shape[1])
        self.device = edge_index.device
        dev = self.device
        E, N = self.E, self.N
        # status words, read back together: [0] range errors, [1]/[2] hub chunks by destination / source, [4] undirected, [5] unmatched
        pair_build = os.environ.get("GSAT_CSR_PAIR", "1") != "0"
        # gsat_build_csr_pair zeroes words [0, 4) itself; [4, 6) are zeroed by the reverse-permutation call that owns them
        self._err = (torch.empty if pair_build else torch.zeros)(8, dtype=torch.int32, device=dev)
        self._err3_fresh = True
        # Both CSRs -- by destination (the forward aggregation order) and by source (the transposed structure of every
        # backward) -- their long-row (hub) chunk lists (empty for molecule-like graphs, built without a host sync), the
        # by-source-slot -> by-destination-slot map and int32 copies of the two edge_index rows: one library call, one sort.
        self.rowptr_dst = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.rowptr_src = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.chunk_ptr_dst = torch.empty(N + 1, dtype=torch.int32, device=dev)
        self.chunk_ptr_src = torch.empty(N + 1, dtype=torch.int32, device=dev)
        ints = torch.empty(7, max((E + 3) // 4 * 4, 4), dtype=torch.int32, device=dev)      # 16-byte aligned rows
        (self.src_by_dst, self.eid_by_dst, self.dst_by_src, self.eid_by_src, self._slot_dst_of_srcslot, self.src32,
         self.dst32) = (ints[i, :E] for i in range(7))
        if os.environ.get("GSAT_CSR_PAIR", "1") == "0":          # A/B switch: the two-sort build through the single-CSR entry points
            src, dst = self.edge_index[0], self.edge_index[1]
            wb = max(call_size("gsat_csr_workspace_bytes", E, N), 256)
            w1 = torch.empty(wb, dtype=torch.uint8, device=dev)
            call("gsat_build_csr", ptr(dst), ptr(src), E, N, ptr(self.rowptr_dst), ptr(self.src_by_dst), ptr(self.eid_by_dst),
                 ptr(self._err), ptr(w1), wb, stream())
            call("gsat_build_csr", ptr(src), ptr(dst), E, N, ptr(self.rowptr_src), ptr(self.dst_by_src), ptr(self.eid_by_src),
                 ptr(self._err), ptr(w1), wb, stream())
            cb = max(call_size("gsat_row_chunks_workspace_bytes", N), 256)
            w2 = torch.empty(cb, dtype=torch.uint8, device=dev)
            call("gsat_row_chunks", ptr(self.rowptr_dst), N, ptr(self.chunk_ptr_dst), ptr(w2), cb, stream())
            call("gsat_row_chunks", ptr(self.rowptr_src), N, ptr(self.chunk_ptr_src), ptr(w2), cb, stream())
            call("gsat_narrow_i64", ptr(src), E, ptr(self.src32), stream())
            call("gsat_narrow_i64", ptr(dst), E, ptr(self.dst32), stream())
            if E:
                inv = torch.empty(E, dtype=torch.int32, device=dev)
                inv[self.eid_by_dst.long()] = torch.arange(E, dtype=torch.int32, device=dev)
                self._slot_dst_of_srcslot.copy_(inv[self.eid_by_src.long()])
            self._partials, self._long, self._tiles = {}, None, {}
            self._checked, self._rev, self._rev_dev, self._rev_flags, self._undirected, self._graphs = False, None, None, None, None, {}
            self._long_async, self._status_queued, self._status_event = None, False, None
            return
        ws_bytes = max(call_size("gsat_csr_pair_workspace_bytes", E, N), 256)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        call("gsat_build_csr_pair", ptr(self.edge_index), E, N, ptr(self.rowptr_dst), ptr(self.src_by_dst), ptr(self.eid_by_dst),
             ptr(self.rowptr_src), ptr(self.dst_by_src), ptr(self.eid_by_src), ptr(self._slot_dst_of_srcslot),
             ptr(self.chunk_ptr_dst), ptr(self.chunk_ptr_src), ptr(self.src32), ptr(self.dst32), ptr(self._err), ptr(ws), ws_bytes,
             stream())
        self._partials = {}
        self._tiles = {}
        self._long = None
        self._checked = False
        self._rev = None
        self._rev_dev = None
        self._rev_flags = None
        self._undirected = None
        self._graphs = {}
        self._long_async = None
        self._status_queued, self._status_event = False, None
        if _SYNC_FREE:
            self._validate_once()

    @property
    def long_rows(self):
        """(by-destination, by-source) chunk lists, or (None, None) when no row has more than GSAT_LONG_ROW_EDGES
        entries -- then the aggregation calls skip the hub-chunk launch.  Costs one small read-back per index
        (per collated batch), merged with the undirected-flag read when that one is needed too."""
        if _SYNC_FREE and self._long is None:
            # inside a captured step nothing is read back: the (early-exit) hub-chunk launches are part of the step iff some batch of
            # this process had a long row (every distinct batch is read back once by the warm-up runs, _validate_once).  Without them
            # a hub row is still summed correctly, by one lane group.
            return (self.chunk_ptr_dst, self.chunk_ptr_src) if _HUBS_SEEN[0] else (None, None)
        if self._long is None or not self._checked:
            self._readback()                     # also for tiny batches: one 32-byte copy answers every host-side question
        return (self.chunk_ptr_dst if self._long[0] else None, self.chunk_ptr_src if self._long[1] else None)

    @property
    def long_rows_nowait(self):
        """`long_rows` without a host sync (see _HUBS_SEEN): exact when this batch's status is already on the host, else the chunk lists
        (whose launches exit early on a batch without hubs) iff hubs have been seen before in this process, else (None, None)."""
        if _STRICT and not _SYNC_FREE and not self._checked and not torch.cuda.is_current_stream_capturing():
            self._readback()                         # strict mode: bad ids / an unsorted batch vector raise HERE, before the first PNA launch
        if self._long is not None:
            known = self._long
        else:
            if not _SYNC_FREE and not torch.cuda.is_current_stream_capturing():
                if not self._status_queued:          # first use of this batch on the sync-free PNA path (its `batch` vector is registered by now)
                    self._status_queued = True
                    _queue_status(self)
                elif self._long_async is None and self._status_event is not None and self._status_event.query():
                    _harvest_status()          # one event query per call while this batch's copy is in flight, the ring only once it landed
            known = self._long_async
        if known is not None:
            return (self.chunk_ptr_dst if known[0] else None, self.chunk_ptr_src if known[1] else None)
        return (self.chunk_ptr_dst, self.chunk_ptr_src) if _HUBS_SEEN[0] else (None, None)

    def partial(self, H: int) -> torch.Tensor:
        """Scratch for the long-row partial sums of width H (upper bound, no host sync); reused across calls
        on the same stream."""
        buf = self._partials.get(H)
        if buf is None:
            n = max(call_size("gsat_long_row_partial_floats", self.E, H), 4)
            buf = torch.empty(n, dtype=torch.float32, device=self.device)
            self._partials[H] = buf
        return buf

    def pna_partial(self, H: int, has_edge_emb: bool) -> torch.Tensor:
        """Record workspace of the PNA kernels' long-row path (upper bound from E, no host sync); reused across calls on the same stream."""
        key = ("pna", H, bool(has_edge_emb))
        buf = self._partials.get(key)
        if buf is None:
            n = max(call_size("gsat_pna_long_row_floats", self.E, H, 1 if has_edge_emb else 0), 4)
            buf = torch.empty(n, dtype=torch.float32, device=self.device)
            self._partials[key] = buf
        return buf

    def _validate_once(self):
        """Sync-free mode never reads the status words inside the step.  The kernels are memory-safe on bad input (ids are
        clamped), but the caller still deserves the ValueError: validate each distinct batch ONCE, outside stream capture
        (the warm-up runs that precede a capture), and never again for the same tensors."""
        # keyed on the tensors' storage AND their versions: a new batch that lands on a recycled allocation of the same shape (or an
        # in-place edit) is validated again
        key = (self.edge_index.data_ptr(), self._key_tensor._version, tuple(self.edge_index.shape), self.N,
               tuple(k for k in self._graphs))
        if key in _VALIDATED or torch.cuda.is_current_stream_capturing():
            return
        self._readback()
        if len(_VALIDATED) > 4096:
            _VALIDATED.clear()
        _VALIDATED.add(key)

    # -- validation (one host sync, deferred until something needs a host-side decision) --------
    def check(self):
        if not self._checked:
            self._readback()

    # -- reverse-edge permutation / undirected flag ----------------------------------------------
    def _build_rev(self):
        E, N, dev = self.E, self.N, self.device
        rev = _i32(E, dev)
        flags = self._err[4:6]                   # zeroed by the call; lives next to the other status words
        if 2 * E <= (1 << 20) and os.environ.get("GSAT_REV_CSR", "1") != "0" and os.environ.get("GSAT_CSR_PAIR", "1") != "0":
            # molecule-to-motif sized batches: pair the copies of (s,d) and (d,s) through the CSRs already built (3 launches, no sort)
            call("gsat_reverse_edge_perm_csr", ptr(self.src32), ptr(self.dst32), ptr(self.rowptr_dst), ptr(self.src_by_dst), ptr(self.eid_by_dst),
                 ptr(self.rowptr_src), ptr(self.dst_by_src), ptr(self.eid_by_src), E, N, ptr(rev), ptr(flags), stream())
        else:
            ws_bytes = max(call_size("gsat_rev_workspace_bytes", E), 256)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            call("gsat_reverse_edge_perm", ptr(self.edge_index), E, N, ptr(rev), ptr(flags), ptr(ws), ws_bytes, stream())
        self._rev_dev, self._rev_flags = rev, flags
        if _SYNC_FREE:
            return
        self._readback()                         # the reference syncs here too (python `if is_undirected`)

    def _readback(self):
        """ONE device->host read for every host-side decision of this batch: undirected flag (if the reverse permutation
        was built), hub-row chunk totals of both CSRs, out-of-range counter."""
        if os.environ.get("GSAT_CSR_PAIR", "1") == "0":
            self._err[1:3] = torch.cat([self.chunk_ptr_dst[-1:], self.chunk_ptr_src[-1:]])
        vals = self._err.tolist()              # every status word of this batch in one copy, no gather kernel
        self._long = (vals[1] > 0, vals[2] > 0)
        if vals[1] > 0 or vals[2] > 0:
            _HUBS_SEEN[0] = True
        # a batch with bad ids stays "unchecked": every later host-side question about this cached index (long_rows, check(), the ops)
        # reads the status words again and raises again, instead of running silently on clamped ids after the first ValueError
        self._checked = vals[0] == 0 and vals[3] == 0
        if vals[0] != 0:
            raise ValueError("edge_index contains node ids outside [0, num_nodes)")
        if vals[3] != 0:
            # the reference's scatter-based InstanceNorm / pools accept any order; PyG collation always yields a sorted vector
            # and the segment pointers here are binary searches over it, so anything else is rejected instead of mis-normalised
            raise ValueError("`batch` must be non-decreasing with ids in [0, num_graphs)")
        if self._rev_flags is not None:
            self._undirected = bool(vals[4])
            self._rev = self._rev_dev if self._undirected else None

    @property
    def rev_and_flag(self):
        """(rev int32[E], flags int32[2]) on the device, no host read: flags[0] = 1 iff the edge set is symmetric."""
        if self._rev_dev is None:
            self._build_rev()
        return self._rev_dev, self._rev_flags

    @property
    def is_undirected(self) -> bool:
        if self._undirected is None:
            if self._rev_dev is None:
                self._build_rev()
            if self._undirected is None:         # built in sync-free mode earlier: read the flag now
                self._readback()
        return self._undirected

    @property
    def rev(self) -> Optional[torch.Tensor]:
        """int32[E] reverse-edge permutation, or None when the edge set is not symmetric."""
        self.is_undirected
        return self._rev

    @property
    def slot_dst_of_srcslot(self) -> torch.Tensor:
        """For slot k of the by-source CSR, the slot of the same edge in the by-destination CSR."""
        return self._slot_dst_of_srcslot

    # -- destination-row windows of the tiled PNA backward -----------------------------------------
    def pna_tiles(self, H: int):
        """(tile_desc int32[T+1,4], T, rows_nominal, rows_cap, edges_cap, spill list) for width H, or None when the tiled backward does not cover H.
        Windows are aligned to graph starts when a batch vector has been registered (graphs()), else fixed."""
        import ctypes
        seg = next(iter(self._graphs.values())) if self._graphs else None
        key = (int(H), id(seg))
        hit = self._tiles.get(key)
        if hit is not None:
            return hit
        if H < 64:            # narrow rows: the two-pass backward is as fast or faster (profiles/r02_summary.md, width table)
            self._tiles[key] = False
            return False
        nominal, slack, ecap = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        budget = int(os.environ.get("GSAT_PNA_TILE_LDS", "0"))
        if _lib.load().gsat_pna_tile_plan(int(H), budget, ctypes.byref(nominal), ctypes.byref(slack), ctypes.byref(ecap)) != 0:
            self._tiles[key] = False
            return False
        T = (self.N + nominal.value - 1) // nominal.value
        tile_ptr = torch.empty(T + 1, 4, dtype=torch.int32, device=self.device)        # (row, rowptr_dst[row], rowptr_src[row], 0) per window
        spill = torch.empty(self.N + 1, dtype=torch.int32, device=self.device)       # [0]: count, [1:]: sources with a spilled edge
        call("gsat_pna_build_tiles", ptr(seg.node_ptr) if seg is not None else None, ptr(seg.node_seg32) if seg is not None else None,
             ptr(self.rowptr_dst), ptr(self.rowptr_src), ptr(self._slot_dst_of_srcslot) if self.E else None, self.N, nominal.value,
             slack.value, ecap.value, ptr(tile_ptr), ptr(spill[1:]), ptr(spill[:1]), stream())
        out = (tile_ptr, T, nominal.value, nominal.value + slack.value, ecap.value, spill)
        self._tiles = {key: out}
        return out

    # -- per-graph segments ----------------------------------------------------------------------
    def graphs(self, batch: torch.Tensor, num_graphs: Optional[int] = None) -> "GraphSegments":
        key = (batch.data_ptr(), batch._version, int(batch.shape[0]))
        seg = self._graphs.get(key)
        if seg is None:
            if not self._err3_fresh:
                self._err[3:4].zero_()             # the order / range counter belongs to the batch vector being registered
            self._err3_fresh = False
            seg = GraphSegments(self, batch, num_graphs)
            self._graphs = {key: seg}
            self._checked = False                  # the next host-side decision re-reads the status words, now including [3]
            if _SYNC_FREE:
                self._validate_once()
        return seg


class GraphSegments:
    """Segment pointers of the ``batch`` vector and of the edges grouped by graph."""

    def __init__(self, index: BatchIndex, batch: torch.Tensor, num_graphs: Optional[int] = None):
        if batch.dtype != torch.int64 or batch.dim() != 1:
            raise ValueError("batch must be an int64 vector")
        self.index = index
        self.batch = batch.contiguous()
        dev = batch.device
        n = int(batch.shape[0])
        if num_graphs is None:
            num_graphs = int(batch.max().item()) + 1 if n > 0 else 0     # reference: batch.max()+1 (sync)
        self.G = int(num_graphs)
        self.node_ptr = torch.empty(self.G + 1, dtype=torch.int32, device=dev)
        flags = index._err[3:4]                   # pre-zeroed status word of the batch index (only ever incremented)
        self.node_seg32 = _i32(n, dev)            # graph id of every node, int32 (written by the same pass that checks the order)
        call("gsat_segment_ptr32", ptr(self.batch), n, self.G, ptr(self.node_ptr), ptr(self.node_seg32), ptr(flags), stream())
        self._flags = flags
        self._edge = None

    def check(self):
        self.index._checked = False
        self.index.check()

    @property
    def edge_segments(self):
        """(edge_ptr int32[G+1], edge_order int32[E], edge_graph int64[E], edge_graph int32[E]): edges grouped by the graph
        of their SOURCE node -- ``batch[col]`` with ``col = edge_index[0]`` (example/gsat.py:133-136)."""
        if self._edge is None:
            ix = self.index
            E, dev = ix.E, ix.device
            eg = torch.empty(max(E, 1), dtype=torch.int64, device=dev)[:E]
            call("gsat_gather_i64", ptr(self.batch), int(self.batch.shape[0]), ptr(ix.edge_index[0].contiguous()), E, ptr(eg), stream())
            ws_bytes = max(call_size("gsat_csr_workspace_bytes", E, self.G), 256)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            eptr = torch.empty(self.G + 1, dtype=torch.int32, device=dev)
            order = _i32(E, dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            call("gsat_build_csr", ptr(eg), None, E, self.G, ptr(eptr), None, ptr(order), ptr(err), ptr(ws), ws_bytes, stream())
            eg32 = _i32(E, dev)
            call("gsat_narrow_i64", ptr(eg), E, ptr(eg32), stream())
            self._edge = (eptr, order, eg, eg32)
        return self._edge


def call_size(name, *args) -> int:
    return int(getattr(_lib.load(), name)(*args))


# ---------------------------------------------------------------------------------------------
# cache: the reference passes the same `data.edge_index` tensor to get_emb, the extractor and the
# masked forward of one step (example/gsat.py:75-89); keying on (storage pointer, version, shape)
# stays correct if the caller mutates edge_index in place (version bump) or swaps the tensor.
# ---------------------------------------------------------------------------------------------
_CACHE: "OrderedDict[tuple, BatchIndex]" = OrderedDict()
_CACHE_SIZE = 8


def get_index(edge_index: torch.Tensor, num_nodes: int) -> BatchIndex:
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), str(edge_index.device))
    ix = _CACHE.get(key)
    if ix is None:
        ix = BatchIndex(edge_index, num_nodes)
        _CACHE[key] = ix
        while len(_CACHE) > _CACHE_SIZE:
            _CACHE.popitem(last=False)
    else:
        _CACHE.move_to_end(key)
    return ix


def clear_cache():
    _CACHE.clear()
    for slot in _STATUS_RING:          # pending status copies of dropped batches are not reported any more
        slot[2] = None
