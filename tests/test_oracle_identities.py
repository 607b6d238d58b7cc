"""not gpu: analytic identities and independent ATen implementations that anchor the oracle
(there are no reference tests or golden vectors for this path: SURVEY.md section 4 / 8c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import given, settings, strategies as st

from oracle import bookkeeping as bk
from oracle import ops
from tests.graphs import random_batch, shuffle_edges


def test_instance_norm_equals_aten_per_graph():
    ei, batch, N = random_batch(0, 5, 2, 20)
    x = torch.randn(N, 12)
    y = ops.instance_norm(x, batch, 5)
    for g in range(5):
        xs = x[batch == g]
        ref = F.instance_norm(xs.t().unsqueeze(0), eps=1e-5).squeeze(0).t()      # ATen: biased var, eps inside sqrt
        assert torch.allclose(y[batch == g], ref, atol=1e-5)


def test_scatter_matches_aten_scatter_reduce():
    idx = torch.randint(0, 7, (50,))
    src = torch.randn(50, 6)
    ex = idx.view(-1, 1).expand_as(src)
    assert torch.allclose(ops.scatter(src, idx, 9, "sum"), torch.zeros(9, 6).scatter_reduce(0, ex, src, "sum"), atol=1e-6)
    assert torch.allclose(ops.scatter(src, idx, 9, "mean"), torch.zeros(9, 6).scatter_reduce(0, ex, src, "mean", include_self=False), atol=1e-6)
    mn = torch.zeros(9, 6).scatter_reduce(0, ex, src, "amin", include_self=False)
    mx = torch.zeros(9, 6).scatter_reduce(0, ex, src, "amax", include_self=False)
    assert torch.equal(ops.scatter(src, idx, 9, "min"), mn) and torch.equal(ops.scatter(src, idx, 9, "max"), mx)
    assert torch.all(ops.scatter(src, idx, 9, "min")[7:] == 0)                    # empty rows -> 0


def test_minmax_gradient_goes_to_first_arg_only():
    src = torch.tensor([[1.0], [0.0], [0.0], [2.0]], requires_grad=True)       # tie at 0 between slots 1 and 2
    out, arg = ops.scatter_minmax(src, torch.tensor([0, 0, 0, 0]), 1, "min")
    out.sum().backward()
    assert arg.item() == 1 and src.grad.view(-1).tolist() == [0.0, 1.0, 0.0, 0.0]


def test_pna_empty_row_std_and_x_i_identity():
    N, H = 6, 4
    ei = torch.tensor([[0, 1, 2], [1, 0, 0]])
    x = torch.randn(N, H)
    att = torch.rand(3, 1)
    out = ops.pna_aggregate(x, ei, att, ["mean", "min", "max", "std"], ["identity"], {"lin": 1.0, "log": 1.0})
    F_in = 2 * H
    assert torch.all(out[3:, :3 * F_in] == 0)
    assert torch.allclose(out[3:, 3 * F_in:], torch.full((3, F_in), 1e-5 ** 0.5))          # conv_layers.py:215-216
    # x_i third: mean = x_i * mean(att) over in-edges  
    a0 = att[[1, 2], 0]
    assert torch.allclose(out[0, :H], x[0] * a0.mean(), atol=1e-6)
    assert torch.allclose(out[0, F_in:F_in + H], torch.minimum(x[0] * a0.min(), x[0] * a0.max()), atol=1e-6)


def test_symmetrise_lift_info_identities():
    ei, batch, N = random_batch(3, 6, 2, 15)
    ei = shuffle_edges(ei, 1)
    E = ei.shape[1]
    rev = torch.from_numpy(bk.reverse_edge_perm(ei, N))
    att = torch.rand(E, 1)
    s = ops.symmetrise(att, rev)
    assert torch.equal(s, s[rev])
    na = torch.rand(N, 1)
    assert torch.equal(ops.lift_node_att_to_edge_att(na, ei), na[ei[0]] * na[ei[1]])
    assert abs(ops.info_loss(torch.full((10, 1), 0.7), 0.7).item()) < 1e-5
    assert ops.get_r(10, 0.1, 0) == 0.9 and abs(ops.get_r(10, 0.1, 25, final_r=0.7) - 0.7) < 1e-12 and ops.get_r(10, 0.1, 100) == 0.5


def test_reorder_like_matches_reverse_perm_and_raises():
    ei, batch, N = random_batch(4, 4, 3, 12)
    E = ei.shape[1]
    vals = np.arange(E, dtype=np.float32)
    assert np.array_equal(bk.reorder_like(ei.flip(0), ei, vals), vals[bk.reverse_edge_perm(ei, N)])
    bad = ei.clone(); bad[1, 0] = (bad[1, 0] + 1) % N
    with pytest.raises(ValueError):
        bk.reorder_like(bad, ei, vals)
    with pytest.raises(ValueError):
        bk.reverse_edge_perm(torch.tensor([[0, 1], [1, 2]]), 3)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 30), st.integers(0, 80), st.integers(0, 10_000))
def test_bookkeeping_properties(n, e, seed):
    rng = np.random.RandomState(seed)
    ei = rng.randint(0, n, size=(2, e)).astype(np.int64)
    rp, perm = bk.csr_by(ei[1], n)
    assert rp[0] == 0 and rp[-1] == e and np.all(np.diff(rp) >= 0)
    assert np.array_equal(np.sort(perm), np.arange(e))                   # CSR round-trips COO
    assert np.all(np.diff(ei[1][perm]) >= 0)
    for r in range(n):
        seg = perm[rp[r]:rp[r + 1]]
        assert np.all(ei[1][seg] == r) and np.all(np.diff(seg) > 0)       # stable: edge-id order inside a row
    sym = np.concatenate([ei, ei[::-1]], axis=1)
    assert bk.is_undirected(sym, n)
    rev = bk.reverse_edge_perm(sym, n)
    assert np.array_equal(rev[rev], np.arange(2 * e))                     # involution, also with duplicates
    assert np.array_equal(sym[0][rev], sym[1]) and np.array_equal(sym[1][rev], sym[0])
    assert bk.is_undirected(ei, n) == np.array_equal(np.sort(bk.edge_keys(ei, n)), np.sort(bk.edge_keys(ei, n, True)))


def test_lpt_sharding_balanced_and_complete():
    e = np.random.RandomState(0).randint(1, 500, size=97)
    parts = bk.shard_graphs_lpt(e, 4)
    assert sorted(sum(parts, [])) == list(range(97))
    loads = [int(e[p].sum()) for p in parts]
    assert max(loads) - min(loads) <= int(e.max())
