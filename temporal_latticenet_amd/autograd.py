"""Training path: autograd wrappers that keep the HIP kernels in the forward and give
`loss.backward()` (train_ln.py:224-233) its gradients.

  * GatherGemmFn  — forward: the MFMA gather-GEMM; backward: dW = im2row(A)^T dOut and dA = scatter(dOut W^T)
                    through the same tap table (materialised im2row kernel + dense products).
  * Im2RowFn      — the differentiable neighbour gather the AFlow arithmetic is built on (lm:301).
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

__all__ = ["grad_mode", "GatherGemmFn", "gather_gemm", "Im2RowFn", "im2row", "PoolFn", "pointnet_pool",
           "group_norm_relu", "slice_gather", "slice_blend", "gru_cell", "pad_rows"]


def grad_mode():
    return torch.is_grad_enabled()


def _tptr(table):
    return C.c_void_p(table.data_ptr()) if table is not None else None


class GatherGemmFn(torch.autograd.Function):
    """out[M,N] = act( im2row(src, table) @ W + bias + residual ); table [M,9] int32 or None (identity rows)"""

    @staticmethod
    def forward(ctx, M, src, table, weight, bias, residual, w_is_nk, relu):
        src = src.contiguous()
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
        c = src.shape[1]
        if table is None:
            rows = src[: ctx.M] if src.shape[0] >= ctx.M else F.pad(src, (0, 0, 0, ctx.M - src.shape[0]))
        else:
            rows = ops.im2row(src, _tptr(table), ctx.M)                          # [M, 9C]
        w_kn = weight.t() if ctx.w_is_nk else weight                             # [K, N]
        dW = rows.t() @ dout                                                     # [K, N]
        if ctx.w_is_nk:
            dW = dW.t()
        dsrc = None
        if ctx.needs_input_grad[1]:
            drows = dout @ w_kn.t()                                              # [M, K]
            if table is None:
                dsrc = torch.zeros_like(src)
                n = min(src.shape[0], ctx.M)
                dsrc[:n] = drows[:n]
            else:
                dsrc = torch.zeros_like(src)
                tl = table.long()
                for t in range(9):
                    idx = tl[:, t]
                    ok = (idx >= 0) & (idx < src.shape[0])
                    dsrc.index_add_(0, idx[ok], drows[ok, t * c:(t + 1) * c])
        dbias = dout.sum(0) if ctx.has_bias else None
        dres = dout if ctx.has_res else None
        return None, dsrc, None, dW.contiguous(), dbias, dres, None, None


def gather_gemm(M, src, table, weight, bias=None, residual=None, w_is_nk=False, relu=False):
    return GatherGemmFn.apply(M, src, table, weight, bias, residual, w_is_nk, relu)


class Im2RowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, table):
        src = src.contiguous()
        ctx.save_for_backward(table)
        ctx.shape = src.shape
        with torch.no_grad():
            return ops.im2row(src, _tptr(table), table.shape[0])

    @staticmethod
    def backward(ctx, drows):
        (table,) = ctx.saved_tensors
        rows, c = ctx.shape
        dsrc = torch.zeros(ctx.shape, dtype=drows.dtype, device=drows.device)
        tl = table.long()
        for t in range(9):
            idx = tl[:, t]
            ok = (idx >= 0) & (idx < rows)
            dsrc.index_add_(0, idx[ok], drows[ok, t * c:(t + 1) * c])
        return dsrc, None


def im2row(src, table):
    return Im2RowFn.apply(src, table)


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


def slice_gather(lv_b, indices, weights):
    idx = indices.long().reshape(-1, 4)
    w = weights.reshape(-1, 4)
    ok = (idx >= 0).float()
    g = lv_b[idx.clamp(min=0)]
    g = torch.cat([g * w[:, :, None], w[:, :, None]], dim=2) * ok[:, :, None]
    return g.reshape(idx.shape[0], -1)


def slice_blend(lv, indices, weights, delta=None):
    idx = indices.long().reshape(-1, 4)
    w = weights.reshape(-1, 4)
    if delta is not None:
        w = w + delta.reshape(-1, 4)
    ok = (idx >= 0).float()
    return (lv[idx.clamp(min=0)] * (w * ok)[:, :, None]).sum(1)
