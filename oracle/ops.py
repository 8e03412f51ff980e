"""ORACLE (test infrastructure only — never imported by the product path).

PyTorch-CPU eager restatement of the operators on the hot path.  Two kinds of functions live here:

 * restatements of IN-TREE reference code, followed line by line and pinned by the golden vectors in
   tests/golden/ (generated from the reference's own classes by tests/golden/make_golden.py):
     gru_step            <- seq_lattice/lattice_modules.py:53-66   (GRUModule.forward)
     aflow_correlation   <- seq_lattice/lattice_modules.py:282-339 (CustomKernelConvLatticeIm2RowModule.forward)
     aflow_step          <- seq_lattice/lattice_modules.py:207-235 (CrossframeLocalInterpolationModule.forward)
     pointnet_pool       <- seq_lattice/lattice_modules.py:448-530 (PointNetSeqModule.forward, pooling half)
     scatter_max/add     <- torch_scatter 2.0.4 semantics as relied on at lm:485-520

 * restatements of the UN-VENDORED lattice ops (parity unpinned, see oracle/permuto.py header): distribute,
   im2row / conv, group norm over the lattice, coarsen / finefy, slice.  Their spec is DESIGN.md.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import permuto as P


# ------------------------------------------------------------------------------------------
# K1 distribute
# ------------------------------------------------------------------------------------------
def distribute(table, positions, values, sigmas, subtract_mean=True, scale_constant=None):
    """positions [N,3] f32, values [N,vd] f32 -> distributed [4N, 3+vd+1], indices [4N] i32, weights [4N]."""
    positions = np.ascontiguousarray(positions, np.float32)
    n = positions.shape[0]
    scale = P.scale_factors(sigmas, scale_constant)
    rem0, rank, bary = P.simplex(P.elevate(positions, scale))
    keys = P.simplex_keys(rem0, rank).reshape(4 * n, 3)
    indices = table.insert(keys)
    weights = bary[:, :4].reshape(-1).astype(np.float32)
    vd = 0 if values is None else values.shape[1]
    dist = np.zeros((4 * n, 3 + vd + 1), np.float32)
    pos4 = np.repeat(positions, 4, axis=0)
    dist[:, :3] = pos4
    if vd:
        dist[:, 3:3 + vd] = np.repeat(np.asarray(values, np.float32), 4, axis=0)
    dist[:, -1] = weights
    if subtract_mean:
        ok = indices >= 0
        v = table.nr_vertices
        # fixed point (units of 2^-20) summed in int64: exact, so the mean is independent of the summation order
        fixed = np.rint(pos4[ok].astype(np.float64) * 1048576.0).astype(np.int64)
        s = np.zeros((v, 3), np.int64)
        np.add.at(s, indices[ok], fixed)
        c = np.bincount(indices[ok], minlength=v).astype(np.float64)
        mean = ((s.astype(np.float64) / np.maximum(c, 1)[:, None]) * (1.0 / 1048576.0)).astype(np.float32)
        dist[ok, :3] = pos4[ok] - mean[indices[ok]]
    return dist, indices.astype(np.int32), weights


# ------------------------------------------------------------------------------------------
# torch_scatter 2.0.4 (dim=0) as the reference relies on it
# ------------------------------------------------------------------------------------------
def scatter_max(src, index, dim_size=None):
    """returns (out, argmax); empty segment -> 0 / src.size(0); ties -> smallest row."""
    rows, c = src.shape
    v = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.full((v, c), float("-inf"), dtype=src.dtype)
    out = out.scatter_reduce(0, index[:, None].expand(-1, c), src, reduce="amax", include_self=True)
    rowid = torch.arange(rows)[:, None].expand(-1, c)
    is_max = src == out[index]
    cand = torch.where(is_max, rowid, torch.full_like(rowid, rows))
    arg = torch.full((v, c), rows, dtype=torch.long)
    arg = arg.scatter_reduce(0, index[:, None].expand(-1, c), cand, reduce="amin", include_self=True)
    out = torch.where(arg == rows, torch.zeros_like(out), out)
    return out, arg


def scatter_add(src, index, dim_size=None):
    v = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((v,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add(0, index, src)


# ------------------------------------------------------------------------------------------
# K2 PointNet pool, reference lm:448-530 (experiment "none"/default branch)
# ------------------------------------------------------------------------------------------
_POOL_LIB = None


def _pool_lib():
    """oracle/_build/libpool_mlp.so (oracle/csrc/pool_mlp.c, built by oracle/Makefile) or False"""
    global _POOL_LIB
    if _POOL_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpool_mlp.so")
        try:
            lib = ctypes.CDLL(path)
            lib.oracle_linear_fma.restype = None
            lib.oracle_linear_fma.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            _POOL_LIB = lib
        except OSError:
            _POOL_LIB = False
    return _POOL_LIB


def linear_fma(x, w, b, relu):
    """y = x @ w.T + b in the ONE summation order DESIGN.md §3.8 fixes for the PointNet MLP:
    acc = b[o]; for i ascending: acc = fma(x[i], w[o][i], acc) — IEEE fp32 fused multiply-add.  Bit-identical to
    csrc/pool.hip.  C restatement when built; otherwise emulated through float64 (the product of two fp32 numbers is
    exact in fp64; the one extra rounding of the sum can differ from a true fma only on an exact fp32 tie)."""
    x = np.ascontiguousarray(np.asarray(x, np.float32))
    w = np.ascontiguousarray(np.asarray(w, np.float32))
    b = None if b is None else np.ascontiguousarray(np.asarray(b, np.float32))
    rows, cin = x.shape
    cout = w.shape[0]
    lib = _pool_lib()
    if lib:
        y = np.empty((rows, cout), np.float32)
        lib.oracle_linear_fma(x.ctypes.data, rows, cin, w.ctypes.data, None if b is None else b.ctypes.data, cout,
                              1 if relu else 0, y.ctypes.data)
        return y
    y = np.broadcast_to(np.zeros(cout, np.float32) if b is None else b, (rows, cout)).astype(np.float32)
    for i in range(cin):
        y = (x[:, i:i + 1].astype(np.float64) * w[None, :, i].astype(np.float64) + y.astype(np.float64)).astype(np.float32)
    return np.maximum(y, np.float32(0)) if relu else y


def pointnet_pool(distributed, indices, nr_vertices, weights, biases, min_points=4, exact=True):
    """exact=True: the MLP in the pinned fma order (linear_fma) — what the HIP kernel is compared with bit for bit;
    exact=False: torch's F.linear (the eager CPU timing baseline of bench.py);
    exact="grad": for gradient checks — the arg-max rows are SELECTED with the pinned fma order (so they are the rows
    the HIP pool selects), the values that autograd differentiates are F.linear's of those very rows."""
    distributed = torch.as_tensor(distributed)
    indices = torch.as_tensor(indices)
    if exact == "grad":
        with torch.no_grad():
            sel_w = [w.detach() for w in weights]
            sel_b = [None if b is None else b.detach() for b in biases]
            hard = pointnet_pool(distributed, indices, nr_vertices, sel_w, sel_b, min_points, exact=True)
            xs = distributed[:, :distributed.shape[1] - 1]
            for i, (w, b) in enumerate(zip(sel_w, sel_b)):
                xs = torch.from_numpy(linear_fma(xs.numpy(), w.numpy(), None if b is None else b.numpy(),
                                                 i != len(sel_w) - 1))
            il = indices.long().clone()
            il[il < 0] = 0
            _, argmax = scatter_max(xs, il, nr_vertices)
        x = distributed[:, :distributed.shape[1] - 1].to(weights[0].dtype)
        for i, (w, b) in enumerate(zip(weights, biases)):
            x = F.linear(x, w, b)
            if i != len(weights) - 1:
                x = torch.relu(x)
        rows, c = x.shape
        live = (argmax < rows) & (hard[:, :c] != 0)                  # empty / masked vertices carry no gradient
        soft = x[argmax.clamp(max=rows - 1), torch.arange(c)[None, :]]
        out = hard.clone().to(x.dtype)
        out[:, :c] = torch.where(live, soft, out[:, :c])
        return out
    barycentric_weights = distributed[:, -1]                                   # lm:448
    x = distributed[:, :distributed.shape[1] - 1]                              # lm:452
    for i, (w, b) in enumerate(zip(weights, biases)):                          # lm:460-473
        last = i == len(weights) - 1
        if exact:
            x = torch.from_numpy(linear_fma(x.numpy(), w.detach().numpy(), None if b is None else b.detach().numpy(),
                                            not last))
        else:
            x = F.linear(x, w, b)
            if not last:
                x = torch.relu(x)
    indices_long = indices.long()                                              # lm:477
    indices_long[indices_long < 0] = 0                                         # lm:480
    reduced, argmax = scatter_max(x, indices_long, nr_vertices)                # lm:512 (dim_size = V, see DESIGN)
    argmax_clone = argmax.clone()
    argmax_clone[argmax > argmax.shape[0]] = 0                                 # lm:514
    argmax_clone[argmax_clone >= distributed.shape[0]] = 0                     # (reference would raise here)
    ones = torch.ones(indices_long.shape[0])
    nr_points = scatter_add(ones, indices_long, nr_vertices).unsqueeze(1)      # lm:519-521
    bary = torch.index_select(barycentric_weights, 0, argmax_clone.flatten())  # lm:522
    bary = bary.view(argmax.shape[0], argmax_clone.shape[1])                   # lm:524
    reduced = torch.cat((reduced, bary), 1)                                    # lm:525
    if min_points > 0:
        reduced = reduced.masked_fill(nr_points < min_points, 0)               # lm:528-530
    return reduced


# ------------------------------------------------------------------------------------------
# K3/K4 im2row + conv, K7 group norm, coarsen / finefy
# ------------------------------------------------------------------------------------------
def im2row(lv, table, pad_rows_value=None):
    """lv [Vs,C], table [M,9] (indices into lv, -1 missing) -> [M, 9*C]; missing neighbour -> zero row."""
    lv = torch.as_tensor(lv)
    table = torch.as_tensor(np.asarray(table)).long()
    m = table.shape[0]
    safe = table.clamp(min=0)
    if pad_rows_value is not None:
        # rows >= lv.shape[0] read as pad value (hidden-state padding)
        beyond = safe >= lv.shape[0]
        safe = safe.clamp(max=lv.shape[0] - 1)
    g = lv[safe.reshape(-1)].reshape(m, 9, lv.shape[1])
    if pad_rows_value is not None:
        g = torch.where(beyond[:, :, None], torch.full_like(g, pad_rows_value), g)
    g = g * (table >= 0)[:, :, None]
    return g.reshape(m, -1)


CONV_ROWS = 1 << 17   # rows of the materialised im2row per product (a 1M-vertex level at C = 192 would be 6.9 GB at once)


def conv(lv, table, weight, bias=None):
    m = np.asarray(table).shape[0]
    if m <= CONV_ROWS:
        out = im2row(lv, table) @ weight
    else:           # the same product in row blocks (BASELINE config 5: ~1M vertices)
        table = np.asarray(table)
        out = torch.cat([im2row(lv, table[r:r + CONV_ROWS]) @ weight for r in range(0, m, CONV_ROWS)])
    return out + bias if bias is not None else out


def gn_groups(c):
    return 32 if c % 32 == 0 else c // 2


def group_norm(lv, gamma, beta, eps=1e-5):
    """GroupNorm over the lattice: statistics over all vertices x channels of the group ([1,C,V] layout)."""
    c = lv.shape[1]
    return F.group_norm(lv.t().unsqueeze(0), gn_groups(c), gamma, beta, eps).squeeze(0).t()


# ------------------------------------------------------------------------------------------
# K9 GRU fusion, reference lm:53-66
# ------------------------------------------------------------------------------------------
def gru_step(lv, h_lv, sd, prefix=""):
    """one GRUModule.forward; returns (new_lv, new_h).  sd holds GRU.* and hidden_linear.* tensors."""
    if h_lv is None:                                                            # lm:54-56
        return lv.clone(), lv.clone()
    h = F.linear(h_lv, sd[prefix + "hidden_linear.weight"], sd[prefix + "hidden_linear.bias"])   # lm:58
    padded = torch.nn.utils.rnn.pad_sequence([h, lv], padding_value=0.0)        # lm:59
    h = padded[:, 0, :].squeeze()                                               # lm:60
    gi = F.linear(lv, sd[prefix + "GRU.weight_ih"], sd[prefix + "GRU.bias_ih"])
    gh = F.linear(h, sd[prefix + "GRU.weight_hh"], sd[prefix + "GRU.bias_hh"])
    c = lv.shape[1]
    r = torch.sigmoid(gi[:, :c] + gh[:, :c])
    z = torch.sigmoid(gi[:, c:2 * c] + gh[:, c:2 * c])
    n = torch.tanh(gi[:, 2 * c:] + r * gh[:, 2 * c:])
    new = (1 - z) * n + z * h                                                   # lm:62 (GRUCell)
    return new, new.clone()


# ------------------------------------------------------------------------------------------
# K10 AFlow, reference lm:282-339 and lm:207-235
# ------------------------------------------------------------------------------------------
def aflow_correlation(x, h_padded, table, alpha, beta, bias, use_center=True):
    v, c = x.shape
    table_t = torch.as_tensor(np.asarray(table)).long()
    nbrs = im2row(h_padded, table_t).reshape(v, 9, c)                           # lm:301 (zero rows for -1)
    valid = (table_t != -1).float()                                             # lm:318
    d = torch.cdist(nbrs, x.unsqueeze(1), p=2.0).squeeze(2)                     # lm:316
    d = d * valid
    if not use_center:
        d[:, -1] = d[:, -1] * 0.0                                               # lm:319-320
    d = d * 1 / (torch.sum(d, dim=1).unsqueeze(1).repeat_interleave(9, dim=1))  # lm:321
    alpha_t = torch.ones_like(d) * alpha
    w = (alpha_t - torch.min(d, alpha_t)) * beta                                # lm:324
    w = w * valid                                                               # lm:325
    if not use_center:
        w[:, -1] = w[:, -1] * 0.0
    out = torch.sum(nbrs.permute(0, 2, 1) * w.unsqueeze(1).repeat_interleave(c, dim=1), axis=2)   # lm:331
    if bias is not None:
        out = out + bias                                                        # lm:333-334
    return out, w, table_t


def aflow_step(lv, h_lv, table, sd, prefix="", use_center=True):
    if h_lv is None:                                                            # lm:208-209
        return lv, lv.clone(), None
    pad = lv.shape[0] - h_lv.shape[0]
    h_padded = F.pad(h_lv, (0, 0, 0, pad), value=-999999)                       # lm:215
    a, w, _ = aflow_correlation(lv, h_padded, table, sd[prefix + "AFLOW.alpha"], sd[prefix + "AFLOW.beta"],
                                sd.get(prefix + "AFLOW.bias"), use_center)
    cat = torch.cat([a, lv], dim=1)                                             # lm:223
    cat = torch.relu(F.linear(cat, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"]))   # lm:226-227
    new = 0.0 * h_padded + (1.0 - 0.0) * cat                                    # lm:229 (alpha = 0.)
    return new, new.clone(), w


# ------------------------------------------------------------------------------------------
# K8 slice
# ------------------------------------------------------------------------------------------
def slice_gather(lv_b, indices, weights):
    """[n, 4*(cb+1)] rows: for each of the 4 simplex vertices [w * b[idx], w] (zeros when idx < 0)."""
    lv_b = torch.as_tensor(lv_b)
    idx = torch.as_tensor(indices).long().reshape(-1, 4)
    w = torch.as_tensor(weights).reshape(-1, 4)
    ok = (idx >= 0).float()
    g = lv_b[idx.clamp(min=0)]                                                  # [n,4,cb]
    g = torch.cat([g * w[:, :, None], w[:, :, None]], dim=2) * ok[:, :, None]
    return g.reshape(idx.shape[0], -1)


def slice_blend(lv, indices, weights, delta=None):
    lv = torch.as_tensor(lv)
    idx = torch.as_tensor(indices).long().reshape(-1, 4)
    w = torch.as_tensor(weights).reshape(-1, 4)
    if delta is not None:
        w = w + torch.as_tensor(delta).reshape(-1, 4)
    ok = (idx >= 0).float()
    return (lv[idx.clamp(min=0)] * (w * ok)[:, :, None]).sum(1)


def splat(values, indices, weights, nr_vertices):
    values = torch.as_tensor(values)
    idx = torch.as_tensor(indices).long()
    w = torch.as_tensor(weights)
    rows = torch.cat([values.repeat_interleave(4, dim=0), torch.ones(idx.shape[0], 1)], 1) * w[:, None]
    ok = idx >= 0
    out = torch.zeros((nr_vertices, rows.shape[1]), dtype=torch.float64)
    return out.index_add(0, idx[ok], rows[ok].double()).float()
