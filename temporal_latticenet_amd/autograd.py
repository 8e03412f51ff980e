"""Training path: autograd wrappers that keep the HIP kernels in the forward and give
`loss.backward()` (train_ln.py:224-233) its gradients.

  * GatherGemmFn  — forward: the MFMA gather-GEMM; backward on HIP kernels too (csrc/backward.hip): dW by
                    tln_gather_gemm_dw (MFMA tiles over (tap, channels, columns), M in slices added in a fixed order), dA
                    as ANOTHER gather-GEMM — for a level's own neighbour table tap k of row v is tap k^1 of its neighbour,
                    so dA[j] = sum_k dOut[table[j, k^1]] W_k^T: the forward kernel over the same table with the weight
                    blocks of paired taps swapped and transposed; 1x1 products: the forward kernel with the other weight
                    layout.  No materialised im2row, no index_add_ (float atomics): the gradients are the same bits on
                    every run.  Only the cross-level products (coarsen / finefy: their tables are no transposes of each
                    other) and channel counts that are no multiple of 32 keep the torch scatter.
  * Im2RowFn      — the differentiable neighbour gather the AFlow arithmetic is built on (lm:301); its backward is a gather
                    through the paired taps as well.
  * SliceBlendFn / SliceGatherFn — the slice blends with a segment-sum backward over the vertex-sorted row list.
  * PoolFn        — forward: the fused MLP + segment-max kernel; backward: the MLP is re-run only on the rows that
                    won the max (one row per (vertex, channel)), so nothing of size [4N, 64] is ever stored.
  * functional torch forms of the elementwise pieces (GroupNorm-apply, GRU gates, slice blends) whose gradients
    torch derives itself.

Inference never comes through here: with gradients disabled the modules call the fused kernels directly.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import ops

__all__ = ["grad_mode", "torch_backward", "GatherGemmFn", "gather_gemm", "Im2RowFn", "im2row", "SliceBlendFn", "SliceGatherFn",
           "PoolFn", "pointnet_pool",
           "group_norm_relu", "slice_gather", "slice_blend", "gru_cell", "pad_rows"]


_PAIR = [1, 0, 3, 2, 5, 4, 7, 6, 8]     # tap k of row v is tap _PAIR[k] of the neighbour it points to (centre: itself)
_TORCH_BACKWARD = False               # tests: the round-2 backward (materialised im2row, index_add_) as the cross-check


def torch_backward(on):
    """test switch: route every backward through the torch formulation (materialised im2row + index_add_)"""
    global _TORCH_BACKWARD
    _TORCH_BACKWARD = bool(on)


def grad_mode():
    return torch.is_grad_enabled()


def _tptr(table):
    return C.c_void_p(table.data_ptr()) if table is not None else None


class GatherGemmFn(torch.autograd.Function):
    """out[M,N] = act( im2row(src, table) @ W + bias + residual ); table [M,9] int32 or None (identity rows)"""

    @staticmethod
    def forward(ctx, M, src, table, weight, bias, residual, w_is_nk, relu, symmetric=False, src_lattice=None):
        src = src.contiguous()
        ctx.symmetric = bool(symmetric)
        ctx.src_lattice = src_lattice
        taps = 9 if table is not None else 1
        with torch.no_grad():
            out = ops.gather_gemm(M, weight, ops.gemm_src(src, _tptr(table), taps), w_is_nk=w_is_nk, bias=bias,
                                  residual=residual, relu=relu)
        ctx.M, ctx.w_is_nk, ctx.relu, ctx.taps = M, w_is_nk, relu, taps
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(src, table, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        src, table, weight, out = ctx.saved_tensors
        dout = dout.contiguous()
        if ctx.relu:
            dout = dout * (out > 0)
        c, n, M, taps = src.shape[1], dout.shape[1], ctx.M, ctx.taps
        hip = c % 32 == 0 and not _TORCH_BACKWARD
        # ---- dW
        if hip:
            dW = ops.gather_gemm_dw(src, _tptr(table), taps, dout, M)               # [taps*c, N], deterministic
        else:
            if table is None:
                rows = src[:M] if src.shape[0] >= M else F.pad(src, (0, 0, 0, M - src.shape[0]))
            else:
                rows = ops.im2row(src, _tptr(table), M)                              # [M, 9C]
            dW = rows.t() @ dout
        if ctx.w_is_nk:
            dW = dW.t()
        # ---- dA
        dsrc = None
        if ctx.needs_input_grad[1]:
            if table is None and hip:
                # dsrc = dout @ W^T: the forward kernel, the same weight tensor read in the other layout
                d = ops.gather_gemm(M, weight, ops.gemm_src(dout, None, 1), w_is_nk=not ctx.w_is_nk)
                if src.shape[0] == M:
                    dsrc = d
                else:
                    dsrc = torch.zeros_like(src)
                    k = min(src.shape[0], M)
                    dsrc[:k] = d[:k]
            elif table is not None and hip and ctx.symmetric and src.shape[0] == M and not ctx.w_is_nk:
                # paired taps: dsrc[j] = sum_k dout[table[j, k^1]] @ W_k^T  -> weight blocks swapped pairwise, transposed
                wp = weight.view(9, c, n)[_PAIR].transpose(1, 2).reshape(9 * n, c).contiguous()
                dsrc = ops.gather_gemm(M, wp, ops.gemm_src(dout, _tptr(table), 9))
            elif table is not None and hip and ctx.src_lattice is not None and not ctx.w_is_nk and \
                    ctx.src_lattice.nr_lattice_vertices() >= src.shape[0]:
                # a cross-level table (coarsen / finefy: no paired taps): drows = dout @ W^T on the forward kernel, then
                # every source row sums the entries that point at it in a FIXED order — the flat table sorted stably by
                # source row (the source level's tln_build_csr), one wave per source row (tln_slice_blend_bwd_lv)
                drows = ops.gather_gemm(M, weight, ops.gemm_src(dout, None, 1), w_is_nk=True)      # [M, 9c]
                flat = table.reshape(-1).contiguous()
                ones = torch.ones((flat.shape[0],), dtype=torch.float32, device=dout.device)
                # (the level may have grown since this frame's forward — backward runs after the whole sequence —: the
                # numbering is append-only, the rows this frame knew come first)
                dsrc = ops.slice_blend_bwd_lv(ctx.src_lattice, drows.view(-1, c), c, ones, None, flat,
                                              per_row=True)[: src.shape[0]].contiguous()
            else:
                w_kn = weight.t() if ctx.w_is_nk else weight                         # [K, N]
                drows = dout @ w_kn.t()                                              # [M, K]
                dsrc = torch.zeros_like(src)
                if table is None:
                    k = min(src.shape[0], M)
                    dsrc[:k] = drows[:k]
                else:
                    tl = table.long()
                    for t in range(9):
                        idx = tl[:, t]
                        ok = (idx >= 0) & (idx < src.shape[0])
                        dsrc.index_add_(0, idx[ok], drows[ok, t * c:(t + 1) * c])
        dbias = dout.sum(0) if ctx.has_bias else None
        dres = dout if ctx.has_res else None
        return None, dsrc, None, dW.contiguous(), dbias, dres, None, None, None, None


def gather_gemm(M, src, table, weight, bias=None, residual=None, w_is_nk=False, relu=False, symmetric=False,
                src_lattice=None):
    """symmetric: `table` is the neighbour table of the level `src` lives on (ConvLatticeModule): its taps pair up, which
    turns dA into a gather-GEMM; the cross-level tables of coarsen / finefy do not — for those `src_lattice` (the level `src`
    lives on) lends its row sorter, so that their dA is a fixed-order segment sum instead of an index_add_"""
    return GatherGemmFn.apply(M, src, table, weight, bias, residual, w_is_nk, relu, symmetric, src_lattice)


class Im2RowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, table, symmetric=False):
        src = src.contiguous()
        ctx.save_for_backward(table)
        ctx.shape = src.shape
        ctx.symmetric = bool(symmetric)
        with torch.no_grad():
            return ops.im2row(src, _tptr(table), table.shape[0])

    @staticmethod
    def backward(ctx, drows):
        (table,) = ctx.saved_tensors
        rows, c = ctx.shape
        m = table.shape[0]
        tl = table.long()
        if ctx.symmetric and rows == m and not _TORCH_BACKWARD:
            # the level's own neighbour table: dsrc[j] = sum_k drows[table[j, k^1], k] — nine gathers, nothing scattered
            d = drows.reshape(m, 9, c)
            dsrc = torch.zeros(ctx.shape, dtype=drows.dtype, device=drows.device)
            for k in range(9):
                idx = tl[:, _PAIR[k]]
                dsrc += torch.where((idx >= 0)[:, None], d[idx.clamp(min=0), k], torch.zeros((), device=drows.device))
            return dsrc, None, None
        dsrc = torch.zeros(ctx.shape, dtype=drows.dtype, device=drows.device)
        for t in range(9):
            idx = tl[:, t]
            ok = (idx >= 0) & (idx < rows)
            dsrc.index_add_(0, idx[ok], drows[ok, t * c:(t + 1) * c])
        return dsrc, None, None


def im2row(src, table, symmetric=False):
    return Im2RowFn.apply(src, table, symmetric)


class PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lattice, distributed, indices, min_points, *params):
        nl = len(params) // 2
        ws, bs = list(params[:nl]), list(params[nl:])
        with torch.no_grad():
            out, argrow = ops.pointnet_pool(lattice, distributed, indices, ws, bs, min_points, want_argrow=True)
        ctx.nl = nl
        ctx.save_for_backward(distributed, argrow, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        distributed, argrow = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        nl = ctx.nl
        if nl == 0:
            return (None, None, None, None)
        cout = argrow.shape[1]
        d = dout[:, :cout]
        mask = argrow >= 0
        rows = argrow[mask].long()
        cols = mask.nonzero()[:, 1]
        g = d[mask]
        ws, bs = [p.detach() for p in params[:nl]], [p.detach() for p in params[nl:]]
        if _TORCH_BACKWARD:
            with torch.enable_grad():
                ps = [p.detach().requires_grad_(True) for p in params]
                x = distributed[rows, : ps[0].shape[1]]
                for i in range(nl):
                    x = F.linear(x, ps[i], ps[nl + i])
                    if i < nl - 1:
                        x = torch.relu(x)
                sel = x.gather(1, cols[:, None]).squeeze(1)
                grads = torch.autograd.grad(sel, ps, g)
            return (None, None, None, None) + tuple(grads)
        # the MLP again on the arg-max rows (one per (vertex, channel)), then its backward layer by layer on the
        # deterministic product kernels: dW_i = h_{i-1}^T dz_i (tln_gather_gemm_dw), dh_{i-1} = dz_i W_i (the forward
        # kernel) — torch's own GEMM may cut the long dimension (R ~ 64 V rows) into atomically added slices
        R = rows.shape[0]
        if R == 0:
            return (None, None, None, None) + tuple(torch.zeros_like(p) for p in params)
        hs = [distributed[rows, : ws[0].shape[1]].contiguous()]
        for i in range(nl):
            y = F.linear(hs[-1], ws[i], bs[i])
            hs.append(torch.relu(y) if i < nl - 1 else y)
        dz = torch.zeros((R, cout), dtype=torch.float32, device=g.device)
        dz[torch.arange(R, device=g.device), cols] = g                      # one entry per row: no collisions
        dws, dbs = [None] * nl, [None] * nl
        for i in range(nl - 1, -1, -1):
            dws[i] = ops.gather_gemm_dw(hs[i], None, 1, dz, R).t().contiguous()      # [cout_i, cin_i]
            dbs[i] = dz.sum(0)
            if i > 0:
                dh = ops.gather_gemm(R, ws[i], ops.gemm_src(dz, None, 1), w_is_nk=False)   # dz @ W_i  ([K=cout_i, N=cin_i])
                dz = dh * (hs[i] > 0)
        return (None, None, None, None) + tuple(dws) + tuple(dbs)


def pointnet_pool(lattice, distributed, indices, weights, biases, min_points):
    return PoolFn.apply(lattice, distributed, indices, min_points, *weights, *biases)


# ---- functional pieces whose gradients torch derives ---------------------------------------------------------
def group_norm_relu(lv, norm, relu=True):
    """GroupNorm over the lattice ([1,C,V] layout) with torch's own differentiable kernel"""
    y = F.group_norm(lv.t().unsqueeze(0), norm.num_groups, norm.weight, norm.bias, norm.eps).squeeze(0).t()
    return torch.relu(y) if relu else y


def pad_rows(h, rows, value=0.0):
    return F.pad(h, (0, 0, 0, rows - h.shape[0]), value=value) if h.shape[0] < rows else h


def gru_cell(x, h_padded, cell):
    gi = F.linear(x, cell.weight_ih, cell.bias_ih)
    gh = F.linear(h_padded, cell.weight_hh, cell.bias_hh)
    c = x.shape[1]
    r = torch.sigmoid(gi[:, :c] + gh[:, :c])
    z = torch.sigmoid(gi[:, c:2 * c] + gh[:, c:2 * c])
    n = torch.tanh(gi[:, 2 * c:] + r * gh[:, 2 * c:])
    return (1 - z) * n + z * h_padded


class SliceGatherFn(torch.autograd.Function):
    """[N, 4*(cb+1)] = for r: [w_r * b[idx_r], w_r]  (tln_slice_gather); backward: segment sum over the vertex-sorted rows"""

    @staticmethod
    def forward(ctx, lattice, lv_b, indices, weights):
        ctx.lattice = lattice
        ctx.save_for_backward(indices, weights)
        ctx.cb = lv_b.shape[1]
        with torch.no_grad():
            return ops.slice_gather(lv_b, indices, weights)

    @staticmethod
    def backward(ctx, dg):
        indices, weights = ctx.saved_tensors
        cb = ctx.cb
        dg = dg.contiguous().reshape(-1, cb + 1)               # row 4p + r holds the gradient of [w_r * b[idx_r], w_r]
        db = ops.slice_blend_bwd_lv(ctx.lattice, dg, cb, weights, None, indices, per_row=True)
        return None, db, None, None


class SliceBlendFn(torch.autograd.Function):
    """[N, C] = sum_r (w_r + delta_r) * lv[idx_r]  (tln_slice); backward: d_lv as a segment sum over the vertex-sorted rows,
    d_delta as one dot product per row — no [N, 4, C] temporaries, no float atomics"""

    @staticmethod
    def forward(ctx, lattice, lv, indices, weights, delta):
        ctx.lattice = lattice
        ctx.has_delta = delta is not None
        ctx.save_for_backward(lv, indices, weights, delta)
        with torch.no_grad():
            return ops.slice_blend(lv, indices, weights, None if delta is None else delta.reshape(-1))

    @staticmethod
    def backward(ctx, dout):
        lv, indices, weights, delta = ctx.saved_tensors
        dout = dout.contiguous()
        dlv = None
        if ctx.needs_input_grad[1]:
            dlv = ops.slice_blend_bwd_lv(ctx.lattice, dout, lv.shape[1], weights,
                                         None if delta is None else delta.reshape(-1), indices)
        ddelta = None
        if ctx.has_delta and ctx.needs_input_grad[4]:
            ddelta = ops.slice_blend_bwd_w(lv, indices, dout).reshape(delta.shape)
        return None, dlv, None, None, ddelta


def slice_gather(lv_b, indices, weights, lattice=None):
    if lattice is not None and not _TORCH_BACKWARD:
        return SliceGatherFn.apply(lattice, lv_b, indices, weights)
    idx = indices.long().reshape(-1, 4)
    w = weights.reshape(-1, 4)
    ok = (idx >= 0).float()
    g = lv_b[idx.clamp(min=0)]
    g = torch.cat([g * w[:, :, None], w[:, :, None]], dim=2) * ok[:, :, None]
    return g.reshape(idx.shape[0], -1)


def slice_blend(lv, indices, weights, delta=None, lattice=None):
    if lattice is not None and not _TORCH_BACKWARD:
        return SliceBlendFn.apply(lattice, lv, indices, weights, delta)
    idx = indices.long().reshape(-1, 4)
    w = weights.reshape(-1, 4)
    if delta is not None:
        w = w + delta.reshape(-1, 4)
    ok = (idx >= 0).float()
    return (lv[idx.clamp(min=0)] * (w * ok)[:, :, None]).sum(1)
