"""Functional wrappers over the C ABI (include/tln.h).  PyTorch supplies device memory and the stream;
every op here launches hand-written gfx950 kernels and raises if the library is unavailable."""
import ctypes as C

import torch

from . import _lib
from . import options as _options
from .lattice import Lattice, stream_ptr, _ptr

__all__ = ["gemm_src", "gather_gemm", "groupnorm_stats", "affine_act", "pointnet_pool", "gru_cell", "aflow",
           "slice_gather", "slice_blend", "slice_deform", "splat", "im2row", "scatter_max", "scatter_add"]

# ---- optional per-call timing (bench.py roofline pass): HIP events on the launch stream ----------------
_prof = None


def profile_begin():
    global _prof
    _prof = []


def profile_end():
    """returns [(name, ms, meta)] for every recorded call; synchronises the device"""
    global _prof
    rec, _prof = _prof, None
    torch.cuda.synchronize()
    return [(n, e0.elapsed_time(e1), meta) for n, e0, e1, meta in (rec or [])]


class _timed:
    def __init__(self, name, **meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        if _prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if _prof is not None:
            self.e1.record()
            _prof.append((self.name, self.e0, self.e1, self.meta))
        return False


_F32 = torch.float32


def _f32c(t):
    if t is None:
        return None
    if t.dtype is _F32 and t.is_cuda and t.is_contiguous():     # the common case, no further work
        return t
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    if not t.is_cuda:
        raise _lib.TlnError("tensor must live on the HIP device")
    return t


def gemm_src(src, table_ptr=None, taps=1, src_rows=None, pad_value=0.0, scale=None, shift=None, relu=False):
    """Describes one A-operand source of gather_gemm; returns (struct, keepalive)."""
    src = _f32c(src)
    s = _lib.GemmSrc()
    s.d_src = src.data_ptr()
    s.src_rows = src.shape[0] if src_rows is None else int(src_rows)
    s.ld = src.stride(0)
    s.cin = src.shape[1]
    s.taps = taps
    s.d_table = table_ptr.value if table_ptr is not None else None
    s.pad_value = float(pad_value)
    scale, shift = _f32c(scale), _f32c(shift)
    s.d_scale = scale.data_ptr() if scale is not None else None
    s.d_shift = shift.data_ptr() if shift is not None else None
    s.relu = 1 if relu else 0
    return s, (src, scale, shift)


def gather_gemm(M, weight, s0, s1=None, w_is_nk=False, bias=None, residual=None, relu=False, out=None, stats=False,
                gn=None):
    """out[M,N] = epi( [gather(s0) | gather(s1)] @ W ).  weight: [K,N] (w_is_nk False) or [N,K].
    stats=True additionally produces the per-32-row (sum, sumsq) of every output column (GroupNorm statistics of the
    next layer) and attaches them to the result as `out._tln_stats`.
    gn=(x, groupnorm_module, relu): GroupNorm(x) (+ReLU) is applied to source 0 inside the same host call; its
    statistics come from x._tln_stats when the producer left them, else from two passes over x."""
    weight = _f32c(weight)
    N = weight.shape[0] if w_is_nk else weight.shape[1]
    K = weight.shape[1] if w_is_nk else weight.shape[0]
    k_expected = s0[0].taps * s0[0].cin + (s1[0].taps * s1[0].cin if s1 is not None else 0)
    if K != k_expected:
        raise _lib.TlnError("gather_gemm: weight K=%d but sources give K=%d" % (K, k_expected))
    bias, residual = _f32c(bias), _f32c(residual)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=weight.device)
    if residual is not None and tuple(residual.shape) != (M, N):
        raise _lib.TlnError("gather_gemm: residual shape %s != (%d,%d)" % (tuple(residual.shape), M, N))
    if bias is not None and bias.numel() != N:
        raise _lib.TlnError("gather_gemm: bias has %d entries, N=%d" % (bias.numel(), N))
    st = torch.empty(((M + 31) // 32, N, 2), dtype=torch.float64, device=weight.device) if (stats and M > 0) else None
    if gn is not None:
        x, norm, gn_relu = gn
        V, Cn = x.shape
        d = _lib.GnDesc()
        part = getattr(x, "_tln_stats", None)
        # scale/shift scratch for the cases the in-kernel finalise does not take (partial sums too large to be re-read
        # by every block, or a shape without 16-byte rows: the library decides)
        keep = torch.empty((2, Cn), dtype=torch.float32, device=weight.device)
        if part is not None and tuple(part.shape) == ((V + 31) // 32, Cn, 2):
            d.d_partials = part.data_ptr()
        else:
            d.ws_bytes = int(_lib.lib().tln_groupnorm_ws_bytes(V, Cn))
            ws = torch.empty((max(d.ws_bytes, 16) // 8,), dtype=torch.float64, device=weight.device)
            d.d_ws, d.d_x = ws.data_ptr(), x.data_ptr()
        d.V, d.C, d.groups, d.relu, d.eps = V, Cn, norm.num_groups, 1 if gn_relu else 0, float(norm.eps)
        d.d_gamma, d.d_beta = norm.weight.data_ptr(), norm.bias.data_ptr()
        d.d_scale_shift = keep.data_ptr() if keep is not None else None
        with _timed("gather_gemm", M=M, N=N, K=K, taps=s0[0].taps, cin=s0[0].cin, res=residual is not None):
            rc = _lib.lib().tln_gn_gather_gemm_opt(C.byref(d), M, N, C.byref(s0[0]),
                                                   C.byref(s1[0]) if s1 is not None else None, _ptr(weight),
                                                   1 if w_is_nk else 0, _ptr(bias), _ptr(residual),
                                                   residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                                                   _ptr(out), out.stride(0), _ptr(st), _options.current_ref(),
                                                   stream_ptr())
        _lib.check(rc, "tln_gn_gather_gemm")
        if st is not None:
            out._tln_stats = st
        return out
    with _timed("gather_gemm", M=M, N=N, K=K, taps=s0[0].taps, cin=s0[0].cin, res=residual is not None):
        opt = _options.current_ref()
        if opt is None:
            rc = _lib.lib().tln_gather_gemm_ex(M, N, C.byref(s0[0]), C.byref(s1[0]) if s1 is not None else None,
                                               _ptr(weight), 1 if w_is_nk else 0, _ptr(bias), _ptr(residual),
                                               residual.stride(0) if residual is not None else 0, 1 if relu else 0,
                                               _ptr(out), out.stride(0), _ptr(st), stream_ptr())
        else:       # the same product under the kernel-selection options in force on this host thread (options.py)
            call = _lib.GemmCall()
            call.M, call.N, call.s0, call.s1 = M, N, C.pointer(s0[0]), (C.pointer(s1[0]) if s1 is not None else None)
            call.d_w, call.w_is_nk, call.d_bias, call.d_residual = _ptr(weight), 1 if w_is_nk else 0, _ptr(bias), _ptr(residual)
            call.ld_res, call.relu = residual.stride(0) if residual is not None else 0, 1 if relu else 0
            call.d_out, call.ld_out, call.d_stats = _ptr(out), out.stride(0), _ptr(st)
            rc = _lib.lib().tln_gather_gemm_opt(C.byref(call), opt, stream_ptr())
    _lib.check(rc, "tln_gather_gemm")
    if st is not None:
        out._tln_stats = st
    return out


def groupnorm_stats(x, groups, gamma, beta, eps=1e-5):
    """per-channel (scale, shift) such that x*scale+shift == GroupNorm_over_all_vertices(x)*gamma+beta.
    If x was produced by gather_gemm(..., stats=True) its partial sums are reused (one tiny launch)."""
    x = _f32c(x)
    V, Cn = x.shape
    ss = torch.empty((2, Cn), dtype=torch.float32, device=x.device)
    scale, shift = ss[0], ss[1]
    st = getattr(x, "_tln_stats", None)
    if st is not None and tuple(st.shape) == ((V + 31) // 32, Cn, 2):
        gamma, beta = _f32c(gamma), _f32c(beta)
        _lib.check(_lib.lib().tln_groupnorm_from_partials(_ptr(st), V, Cn, groups, _ptr(gamma), _ptr(beta),
                                                          float(eps), _ptr(scale), _ptr(shift), stream_ptr()),
                   "tln_groupnorm_from_partials")
        return scale, shift
    ws_bytes = int(_lib.lib().tln_groupnorm_ws_bytes(V, Cn))
    ws = torch.empty((max(ws_bytes, 16) // 8,), dtype=torch.float64, device="cuda")
    gamma, beta = _f32c(gamma), _f32c(beta)
    _lib.check(_lib.lib().tln_groupnorm_stats(_ptr(x), V, Cn, groups, _ptr(gamma), _ptr(beta), float(eps),
                                              _ptr(scale), _ptr(shift), _ptr(ws), ws_bytes, stream_ptr()),
               "tln_groupnorm_stats")
    return scale, shift


def affine_act(x, scale, shift, relu=False):
    x = _f32c(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().tln_affine_act(_ptr(x), x.shape[0], x.shape[1], _ptr(_f32c(scale)), _ptr(_f32c(shift)),
                                         1 if relu else 0, _ptr(out), stream_ptr()), "tln_affine_act")
    return out


def pointnet_pool(lattice: Lattice, distributed, indices, weights, biases, min_points=4, want_argrow=False):
    """Fused per-row MLP + segment max (+argmax barycentric) of PointNetSeqModule (lm:448-530).
    want_argrow=True also returns [V, cout] int32: the row that produced each pooled value (-1 = empty/masked)."""
    # rows straight from this lattice's last distribute: the native pool takes them from the vertex bins (it checks the
    # buffer address); anything else (other tensors, edited rows) goes through a vertex-sorted row list
    from_bins = lattice.bins_describe(distributed, indices)
    distributed = _f32c(distributed)
    if not from_bins:
        if lattice.bins_valid_for(distributed):
            lattice.drop_bins()
        lattice.ensure_csr(indices)
    rows, cols = distributed.shape
    nl = len(weights)
    ws = [_f32c(w) for w in weights]
    bs = [_f32c(b) for b in biases]
    dims = [ws[0].shape[1]] + [w.shape[0] for w in ws] if nl else [cols - 1]
    V = lattice.nr_lattice_vertices()
    out = torch.empty((V, 2 * dims[-1]), dtype=torch.float32, device="cuda")
    warr = (C.c_void_p * max(nl, 1))(*[w.data_ptr() for w in ws])
    barr = (C.c_void_p * max(nl, 1))(*[b.data_ptr() for b in bs])
    darr = (C.c_int * (nl + 1))(*dims)
    argrow = torch.empty((V, dims[-1]), dtype=torch.int32, device="cuda") if want_argrow else None
    lattice.apply_options()
    with _timed("pointnet_pool", rows=rows, V=V, cout=dims[-1]):
        rc = _lib.lib().tln_pointnet_pool_ex(lattice._h, _ptr(distributed), rows, cols, nl, warr, barr, darr,
                                             int(min_points), _ptr(out), _ptr(argrow), stream_ptr())
    _lib.check(rc, "tln_pointnet_pool")
    return (out, argrow) if want_argrow else out


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """GRUCell(x, pad(h)) with h zero-padded to x.shape[0] rows (lm:59-62)."""
    x, h = _f32c(x), _f32c(h)
    V, Cn = x.shape
    out = torch.empty_like(x)
    ws = torch.empty((V * 6 * Cn,), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_gru_cell_opt(_ptr(x), _ptr(h), V, h.shape[0], Cn, _ptr(_f32c(w_ih)), _ptr(_f32c(w_hh)),
                                           _ptr(_f32c(b_ih)), _ptr(_f32c(b_hh)), _ptr(out), _ptr(ws), ws.numel(),
                                           _options.current_ref(), stream_ptr()), "tln_gru_cell")
    return out


def lstm_gates(gates, C):
    gates = _f32c(gates)
    V = gates.shape[0]
    out = torch.empty((V, C), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_lstm_gates(_ptr(gates), V, C, _ptr(out), stream_ptr()), "tln_lstm_gates")
    return out


def temporal_max(x, h, pad_value=-9999.0):
    x, h = _f32c(x), _f32c(h)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().tln_temporal_max(_ptr(x), _ptr(h), x.shape[0], h.shape[0], x.shape[1], float(pad_value),
                                           _ptr(out), stream_ptr()), "tln_temporal_max")
    return out


def cga_gate(a, x, Vh, scale):
    a, x = _f32c(a), _f32c(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().tln_cga_gate(_ptr(a), _ptr(x), x.shape[0], int(Vh), x.shape[1], float(scale), _ptr(out),
                                       stream_ptr()), "tln_cga_gate")
    return out


def fill_empty_rows(x, half, value):
    x = _f32c(x)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().tln_fill_empty_rows(_ptr(x), x.shape[0], x.shape[1], int(half), float(value), _ptr(out),
                                              stream_ptr()), "tln_fill_empty_rows")
    return out


def aflow(x, h, table_ptr, alpha, beta, bias=None, pad_value=-999999.0, use_center=True):
    x, h = _f32c(x), _f32c(h)
    V, Cn = x.shape
    out = torch.empty_like(x)
    w = torch.empty((V, 9), dtype=torch.float32, device="cuda")
    idx = torch.empty((V, 9), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().tln_aflow(_ptr(x), _ptr(h), V, h.shape[0], Cn, table_ptr, float(alpha), float(beta),
                                    float(pad_value), 1 if use_center else 0, _ptr(_f32c(bias)), _ptr(out), _ptr(w),
                                    _ptr(idx), stream_ptr()), "tln_aflow")
    return out, w, idx


def slice_gather(lv, indices, weights):
    lv = _f32c(lv)
    n = indices.shape[0] // 4
    cb = lv.shape[1]
    out = torch.empty((n, 4 * (cb + 1)), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_slice_gather(_ptr(lv), lv.shape[0], cb, _ptr(indices.contiguous()),
                                           _ptr(_f32c(weights)), n, _ptr(out), stream_ptr()), "tln_slice_gather")
    return out


def slice_blend(lv, indices, weights, delta=None, bias=None):
    lv = _f32c(lv)
    n = indices.shape[0] // 4
    out = torch.empty((n, lv.shape[1]), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_slice(_ptr(lv), lv.shape[0], lv.shape[1], _ptr(indices.contiguous()),
                                    _ptr(_f32c(weights)), _ptr(_f32c(delta)), _ptr(_f32c(bias)), n, _ptr(out),
                                    stream_ptr()), "tln_slice")
    return out


def gather_gemm_dw(src, table_ptr, taps, dout, M):
    """dW [taps*cin, N] of out = im2row(src, table) @ W (tln_gather_gemm_dw: MFMA tiles, slices of M added in a fixed
    order); cin a multiple of 32"""
    src, dout = _f32c(src), _f32c(dout)
    cin, n = src.shape[1], dout.shape[1]
    lib = _lib.lib()
    ws = torch.empty((int(lib.tln_gather_gemm_dw_ws_floats(M, cin, taps, n)),), dtype=torch.float32, device="cuda")
    dw = torch.empty((taps * cin, n), dtype=torch.float32, device="cuda")
    _lib.check(lib.tln_gather_gemm_dw(_ptr(src), src.shape[0], cin, table_ptr if taps > 1 else None, taps, _ptr(dout), M, n,
                                      _ptr(dw), _ptr(ws), ws.numel(), stream_ptr()), "tln_gather_gemm_dw")
    return dw


def slice_blend_bwd_lv(lattice: Lattice, dvals, C, weights, delta, indices, per_row=False):
    """[V, C] = per-vertex sum of (w + delta)_row * dvals[row >> 2 (per_row: row)][:C] over the vertex-sorted row list"""
    lattice.ensure_csr(indices)
    dvals = _f32c(dvals)
    out = torch.empty((lattice.nr_lattice_vertices(), C), dtype=torch.float32, device="cuda")
    rows = indices.shape[0]
    _lib.check(_lib.lib().tln_slice_blend_bwd_lv(lattice._h, _ptr(dvals), dvals.shape[-1], C, 1 if per_row else 0,
                                                 _ptr(_f32c(weights)), _ptr(_f32c(delta)), rows, _ptr(out), stream_ptr()),
               "tln_slice_blend_bwd_lv")
    return out


def slice_blend_bwd_w(lv, indices, dout):
    """[4N] = dot(lv[idx_row], dout[row >> 2])"""
    lv, dout = _f32c(lv), _f32c(dout)
    rows = indices.shape[0]
    dw = torch.empty((rows,), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_slice_blend_bwd_w(_ptr(lv), lv.shape[0], lv.shape[1], _ptr(indices.contiguous()), _ptr(dout),
                                                rows, _ptr(dw), stream_ptr()), "tln_slice_blend_bwd_w")
    return dw


def slice_deform(b, scores, indices, weights, w_pre, w_dw, b_dw, bias=None):
    """the DeformSlice head per point in one kernel (tln_slice_deform): logits [n, C]"""
    b, scores = _f32c(b), _f32c(scores)
    n = indices.shape[0] // 4
    out = torch.empty((n, scores.shape[1]), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_slice_deform(_ptr(b), b.shape[1], _ptr(scores), scores.shape[0], scores.shape[1],
                                           _ptr(indices.contiguous()), _ptr(_f32c(weights)), _ptr(_f32c(w_pre)),
                                           _ptr(_f32c(w_dw)), _ptr(_f32c(b_dw)), _ptr(_f32c(bias)), n, _ptr(out),
                                           stream_ptr()), "tln_slice_deform")
    return out


def splat(lattice: Lattice, values, indices, weights):
    lattice.ensure_csr(indices)
    values = _f32c(values)
    val_dim = values.shape[1] if values is not None else 0
    V = lattice.nr_lattice_vertices()
    out = torch.empty((V, val_dim + 1), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_splat(lattice._h, _ptr(values), val_dim, _ptr(_f32c(weights)), indices.shape[0],
                                    _ptr(out), stream_ptr()), "tln_splat")
    return out


def im2row(src, table_ptr, M):
    src = _f32c(src)
    out = torch.empty((M, 9 * src.shape[1]), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_im2row(_ptr(src), src.shape[0], src.shape[1], table_ptr, M, _ptr(out), stream_ptr()),
               "tln_im2row")
    return out


def scatter_max(src, index, out_rows):
    src = _f32c(src)
    index = index.contiguous().long()
    rows, Cn = src.shape
    out = torch.empty((out_rows, Cn), dtype=torch.float32, device="cuda")
    arg = torch.empty((out_rows, Cn), dtype=torch.int64, device="cuda")
    ws = torch.empty((max(out_rows * Cn, 1),), dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().tln_scatter_max(_ptr(src), _ptr(index), rows, Cn, out_rows, _ptr(out), _ptr(arg), _ptr(ws),
                                          ws.numel() * 8, stream_ptr()), "tln_scatter_max")
    return out, arg


def scatter_add(src, index, out_rows, out=None):
    src = _f32c(src)
    index = index.contiguous().long()
    rows, Cn = src.shape
    if out is None:
        out = torch.zeros((out_rows, Cn), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().tln_scatter_add(_ptr(src), _ptr(index), rows, Cn, out_rows, _ptr(out), stream_ptr()),
               "tln_scatter_add")
    return out
