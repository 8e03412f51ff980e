"""Frame program: LNN_SEQ.forward (reference seq_lattice/models.py:284-476) of one frame as two native calls.

The operator-level route (models.py / lattice_modules.py / seq_modules.py of this package) makes ~150 Python-level
calls per frame; on a lattice of a few thousand vertices that host work is as long as the kernels.  Here the module
tree is walked ONCE, after the lazily created parameters exist, and turned into the op list of include/tln.h
("Frame program"); every later inference frame is `tln_program_begin_frame` + `tln_program_run`.  The ops call the
same C-ABI entry points with the same arguments in the same order as the modules do, so both routes produce
identical tensors (tests/test_gpu_engine.py compares them bit for bit).

Supported: experiment none / slice_no_deform / pointnet_no_local_mean and every fusion choice of `rnn_modules`
(none / gru / aflow / linear / lstm / maxpool / cga) in every slot.  The no-elevation / attention-pool experiments,
gradients, dropout in training mode and the AFlow visualisation hooks stay on the operator-level route
(compile_model returns None).
"""
import ctypes as C

import torch

from . import _lib
from .lattice import stream_ptr
from .lattice_modules import BottleneckBlock, NO_MEAN_EXPERIMENTS, ResnetBlock
from .seq_modules import (CrossframeGlobalAttentionModule, CrossframeLocalInterpolationModule, GRUModule, LSTMModule,
                          TemporalLinearModule, TemporalMaxPoolModule)

ROWS_POINTS, ROWS_POINT_ROWS, ROWS_STATE = -1, -2, -16
MAX_LEVELS = 8          # TLN_MAX_LEVELS (include/tln.h)
SLOT_F32, SLOT_STATS, SLOT_STATE_NEW, SLOT_STATE_PREV, SLOT_OUT = 0, 1, 2, 3, 4
TABLE_NONE, TABLE_NBR, TABLE_C2F, TABLE_F2C = 0, 1, 2, 3
(OP_GEMM, OP_GN_PARTIALS, OP_POOL, OP_GRU, OP_AFLOW, OP_SLICE_GATHER, OP_SLICE, OP_COPY, OP_ZERO_ROW0,
 OP_STOP_IF_EARLY, OP_SLICE_DEFORM, OP_LSTM_GATES, OP_TEMPORAL_MAX, OP_CGA_GATE, OP_FILL_EMPTY) = range(1, 16)


class NotReady(Exception):
    """a lazily created parameter does not exist yet (the module has not seen a frame)"""


class Unsupported(Exception):
    """the configuration needs the operator-level route"""


class Val:
    """a [rows, cols] value of the program: its slot, its level (or row code) and, if a product left them, the
    slot with the GroupNorm partial sums of it"""

    def __init__(self, slot, rows, cols, stats=None):
        self.slot, self.rows, self.cols, self.stats = slot, rows, cols, stats


def _p(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise Unsupported("parameter is not a contiguous fp32 device tensor")
    return t.data_ptr()


def _need(x, what):
    if x is None:
        raise NotReady(what)
    return x


class Builder:
    def __init__(self):
        self.slots, self.ops, self.keep = [], [], []
        self.n_states = 0

    # ---- slots
    def slot(self, rows, cols, kind=SLOT_F32, state=-1):
        self.slots.append((rows, cols, kind, state))
        return len(self.slots) - 1

    def new_state(self, level, cols):
        sid = self.n_states
        self.n_states += 1
        new = self.slot(level, cols, SLOT_STATE_NEW, sid)
        prev = self.slot(level, cols, SLOT_STATE_PREV, sid)
        return sid, new, prev

    # ---- ops
    def op(self, kind, **kw):
        o = dict(kind=kind, cond_state=-1, cond_has=0, out=-1, out_col=0, n=0, stats_out=-1, s0=None, s1=None,
                 residual=-1, relu=0, w_is_nk=0, w=None, bias=None, p=[None] * 8, i=[0] * 8, f=[0.0] * 4)
        o.update(kw)
        self.ops.append(o)
        return o

    @staticmethod
    def src(val, table=TABLE_NONE, level=0, relu=False, pad_value=0.0, gn=None):
        """gn = torch.nn.GroupNorm whose statistics are val.stats"""
        d = dict(slot=val.slot, table=table, level=level, relu=1 if relu else 0, pad_value=float(pad_value), gn_stats=-1,
                 gn_groups=0, gn_eps=0.0, gn_gamma=None, gn_beta=None)
        if gn is not None:
            assert val.stats is not None
            d.update(gn_stats=val.stats, gn_groups=gn.num_groups, gn_eps=float(gn.eps), gn_gamma=_p(gn.weight),
                     gn_beta=_p(gn.bias))
        return d

    def with_stats(self, val):
        """GroupNorm needs the partial sums of val; a product's epilogue leaves them, anything else gets one pass"""
        if val.stats is None:
            val.stats = self.slot(val.rows, val.cols, SLOT_STATS)
            self.op(OP_GN_PARTIALS, s0=self.src(val), stats_out=val.stats)
        return val

    def gemm(self, rows, n, weight, w_is_nk, s0, s1=None, bias=None, residual=None, relu=False, stats=True, cond=None,
             out=None, out_col=0, out_cols=None):
        self.keep.append(weight)
        if out is None:
            out = self.slot(rows, n if out_cols is None else out_cols)
        st = self.slot(rows, n, SLOT_STATS) if stats else -1
        o = self.op(OP_GEMM, out=out, out_col=out_col, n=n, stats_out=st, s0=s0, s1=s1,
                    residual=residual.slot if residual is not None else -1, relu=1 if relu else 0,
                    w_is_nk=1 if w_is_nk else 0, w=_p(weight), bias=_p(bias))
        if cond is not None:
            o["cond_state"], o["cond_has"] = cond
        return Val(out, rows, n if out_cols is None else out_cols, st if stats else None)

    # ---- modules (same order of C-ABI calls as the operator-level forward of each class)
    def tap_product(self, mod, x, rows, table, level, norm=None, residual=None, out=None, out_col=0, out_cols=None):
        """_TapConv._product: [GroupNorm+ReLU prologue ->] 9-tap product, partial sums of the output"""
        w = _need(mod.weight, "conv weight")
        gn = None
        if norm is not None:
            gn = _need(norm.norm, "GroupNorm")
            x = self.with_stats(x)
        return self.gemm(rows, w.shape[1], w, False, self.src(x, table, level, relu=norm is not None, gn=gn),
                         bias=mod.bias, residual=residual, out=out, out_col=out_col, out_cols=out_cols)

    def gn_relu_1x1(self, mod, x, residual=None):
        lin = _need(mod.linear.linear, "1x1 weight")
        gn = _need(mod.norm.norm, "GroupNorm")
        x = self.with_stats(x)
        return self.gemm(x.rows, lin.weight.shape[0], lin.weight, True, self.src(x, relu=True, gn=gn), bias=lin.bias,
                         residual=residual)

    def gn_relu_conv(self, mod, x, level, residual=None):
        if mod.drop is not None and mod.drop.prob > 0.0:
            self.uses_dropout = True
        return self.tap_product(mod.conv, x, level, TABLE_NBR, level, mod.norm, residual)

    def block(self, mod, x, level):
        if isinstance(mod, ResnetBlock):
            t = self.gn_relu_conv(mod.conv1, x, level)
            return self.gn_relu_conv(mod.conv2, t, level, residual=x)
        if isinstance(mod, BottleneckBlock):
            t = self.gn_relu_1x1(mod.contract, x)
            t = self.gn_relu_conv(mod.conv, t, level)
            return self.gn_relu_1x1(mod.expand, t, residual=x)
        raise Unsupported(type(mod).__name__)

    def fusion(self, mod, x, level):
        """GRUModule (lm:42-66) / CrossframeLocalInterpolationModule (lm:188-235); the module's output IS the
        state it stores.  First frame of a sequence: the state is the input itself (lm:54-56, 208-209)."""
        if mod is None:
            return x
        sid, new, prev = self.new_state(level, x.cols)
        prev_val = Val(prev, ROWS_STATE - sid, x.cols)
        self.op(OP_COPY, out=new, s0=self.src(x), cond_state=sid, cond_has=0)
        if isinstance(mod, GRUModule):
            g = mod.GRU
            h1 = self.gemm(ROWS_STATE - sid, x.cols, mod.hidden_linear.weight, True, self.src(prev_val),
                           bias=mod.hidden_linear.bias, stats=False, cond=(sid, 1))        # lm:58
            self.keep += [g.weight_ih, g.weight_hh, g.bias_ih, g.bias_hh]
            self.op(OP_GRU, out=new, s0=self.src(x), s1=self.src(h1), cond_state=sid, cond_has=1,
                    p=[_p(g.weight_ih), _p(g.weight_hh), _p(g.bias_ih), _p(g.bias_hh)] + [None] * 4)   # lm:59-62
        elif isinstance(mod, CrossframeLocalInterpolationModule):
            a = mod.AFLOW
            if a.first_time:
                raise NotReady("AFLOW parameters")
            av = self.slot(level, x.cols)
            self.keep.append(a.bias)
            self.op(OP_AFLOW, out=av, s0=dict(self.src(x), level=level), s1=self.src(prev_val), cond_state=sid,
                    cond_has=1, bias=_p(a.bias) if a.use_bias else None,
                    f=[float(a.alpha), float(a.beta), -999999.0, 0.0], i=[1 if a.use_center else 0] + [0] * 7)
            out = self.gemm(level, x.cols, mod.linear.weight, True, self.src(Val(av, level, x.cols)), self.src(x),
                            bias=mod.linear.bias, relu=True, stats=True, cond=(sid, 1), out=new)  # lm:223-227
            # the product leaves the GroupNorm partial sums of the new state; the first frame (a plain copy) gets
            # them from one pass, into the same slot
            self.op(OP_GN_PARTIALS, s0=self.src(Val(new, level, x.cols)), stats_out=out.stats, cond_state=sid, cond_has=0)
            return Val(new, level, x.cols, out.stats)
        elif isinstance(mod, TemporalLinearModule):
            if x.cols != mod.nr_output_channels:
                raise Unsupported("TemporalLinearModule width")
            h1 = self.gemm(ROWS_STATE - sid, x.cols, mod.hidden_linear.weight, True, self.src(prev_val),
                           bias=mod.hidden_linear.bias, stats=False, cond=(sid, 1))        # lm:172
            out = self.gemm(level, x.cols, mod.linear.weight, True, self.src(h1, pad_value=0.0), self.src(x),
                            bias=mod.linear.bias, relu=True, stats=True, cond=(sid, 1), out=new)  # lm:174-179
            self.op(OP_GN_PARTIALS, s0=self.src(Val(new, level, x.cols)), stats_out=out.stats, cond_state=sid, cond_has=0)
            return Val(new, level, x.cols, out.stats)
        elif isinstance(mod, LSTMModule):
            cell = mod.lstm
            h1 = self.gemm(ROWS_STATE - sid, x.cols, mod.hidden_linear.weight, True, self.src(prev_val),
                           bias=mod.hidden_linear.bias, stats=False, cond=(sid, 1))        # lm:32
            gi = self.gemm(level, 4 * x.cols, cell.weight_ih, True, self.src(x), bias=cell.bias_ih, stats=False,
                           cond=(sid, 1))
            gates = self.gemm(level, 4 * x.cols, cell.weight_hh, True, self.src(h1, pad_value=0.0), bias=cell.bias_hh,
                              residual=gi, stats=False, cond=(sid, 1))                     # lm:33-36
            self.op(OP_LSTM_GATES, out=new, s0=self.src(gates), cond_state=sid, cond_has=1)
        elif isinstance(mod, TemporalMaxPoolModule):
            self.op(OP_TEMPORAL_MAX, out=new, s0=self.src(x), s1=self.src(prev_val), cond_state=sid, cond_has=1,
                    f=[-9999.0, 0.0, 0.0, 0.0])                                            # lm:138-141
        elif isinstance(mod, CrossframeGlobalAttentionModule):
            lin = _need(mod.conv.linear, "attention conv")
            gn = _need(mod.groupnorm.norm, "attention GroupNorm")
            h1 = self.gemm(ROWS_STATE - sid, x.cols, mod.hidden_linear.weight, True, self.src(prev_val),
                           bias=mod.hidden_linear.bias, stats=False, cond=(sid, 1))        # lm:89
            a = self.gemm(level, x.cols, lin.weight, True, self.src(h1, pad_value=0.0), relu=True, stats=True,
                          cond=(sid, 1))                                                   # lm:90-98
            a2 = self.gemm(level, x.cols, lin.weight, True, self.src(a, gn=gn), stats=False, cond=(sid, 1))  # lm:100-102
            self.op(OP_CGA_GATE, out=new, s0=self.src(a2), s1=self.src(x), cond_state=sid, cond_has=1,
                    i=[sid] + [0] * 7)                                                     # lm:104-112
        else:
            raise Unsupported(type(mod).__name__)
        return Val(new, level, x.cols)


def _walk_model(model):
    """mirrors LNN_SEQ.forward (this package's models.py, reference models.py:284-476)"""
    b = Builder()
    b.uses_dropout = False
    seq = model.sequence_learning
    rnn = model.rnn_modules if seq else ["none"] * 4
    for k in rnn:
        if k not in ("none", "gru", "aflow", "linear", "lstm", "maxpool", "cga"):
            raise Unsupported("fusion module " + k)
    pn = model.point_net_seq
    if pn.experiment not in ("none", "slice_no_deform", "pointnet_no_local_mean"):
        raise Unsupported("experiment " + pn.experiment)
    if pn.first_time:
        raise NotReady("PointNet layers")
    if len(pn.layers) > 4:
        raise Unsupported("more than four PointNet layers")

    # ---- PointNetSeqModule.forward (lm:407-576)
    dims = [pn.layers[0].weight.shape[1]] + [l.weight.shape[0] for l in pn.layers]
    pooled = Val(b.slot(0, 2 * dims[-1]), 0, 2 * dims[-1])
    ws = [l.weight for l in pn.layers]
    bs = [l.bias for l in pn.layers]
    b.keep += ws + bs
    early_maxpool = seq and rnn[0] == "maxpool"
    b.op(OP_POOL, out=pooled.slot, p=[_p(w) for w in ws] + [None] * (4 - len(ws)) + [_p(x) for x in bs] +
         [None] * (4 - len(bs)), i=[len(ws)] + dims + [0] * (5 - len(dims)) + [0 if early_maxpool else 4, 0])
    if early_maxpool:                                    # lm:555-562: empty vertices must lose the max
        filled = Val(b.slot(0, pooled.cols), 0, pooled.cols)
        b.op(OP_FILL_EMPTY, out=filled.slot, s0=b.src(pooled), i=[pooled.cols // 2] + [0] * 7, f=[-9900.0, 0.0, 0.0, 0.0])
        pooled = filled
    fm = pn.fusion_module if seq else None
    x = b.fusion(fm, pooled, 0)
    if fm is not None:                                   # the stored state keeps its row 0 (lm:569-570 on a copy)
        q = Val(b.slot(0, x.cols), 0, x.cols)
        b.op(OP_COPY, out=q.slot, s0=b.src(x), i=[1] + [0] * 7)      # clone with row 0 zeroed, one launch
        x = q
    else:
        b.op(OP_ZERO_ROW0, out=x.slot)
    lv = b.tap_product(_need(getattr(pn, "last_conv", None), "last_conv"), x, 0, TABLE_NBR, 0)      # lm:573

    stopped = False

    def stop(v):
        nonlocal stopped
        if not stopped:
            b.op(OP_STOP_IF_EARLY, s0=b.src(v))
            b.stop_shape = (v.rows, v.cols)
            stopped = True

    if seq and rnn[1] == "none" and rnn[2] == "none" and rnn[3] == "none":                          # models:307
        stop(lv)
    fusion = list(model.recurrent_fusion_modules) if seq else [None, None, None]
    skips = []
    level = 0
    for i in range(model.nr_downsamples):                                                            # models:314
        for blk in model.resnet_blocks_per_down_lvl_list[i]:
            lv = b.block(blk, lv, level)
        skips.append((lv, level))
        if i == 0:
            lv = b.fusion(fusion[0], lv, level)                                                      # models:341
            if seq and rnn[2] == "none" and rnn[3] == "none":                                        # models:346
                stop(lv)
        cm = model.coarsens_list[i]
        lv = b.tap_product(cm.coarse, lv, level + 1, TABLE_C2F, level + 1, cm.norm)                  # models:353
        level += 1
    for blk in model.resnet_blocks_bottleneck:                                                       # models:361
        lv = b.block(blk, lv, level)
    lv = b.fusion(fusion[1], lv, level)                                                              # models:381
    i = 0
    for i in range(model.nr_downsamples):                                                            # models:390
        fine, fine_level = skips.pop()
        fm_ = model.finefy_list[i]
        w = _need(fm_.fine.weight, "finefy weight")
        n_f = w.shape[1]
        if model.do_concat_for_vertical_connection:                                                  # models:401
            cat = b.slot(fine_level, n_f + fine.cols)
            b.tap_product(fm_.fine, lv, fine_level, TABLE_F2C, level, fm_.norm, out=cat, out_col=0,
                          out_cols=n_f + fine.cols)                                                  # models:398
            b.op(OP_COPY, out=cat, out_col=n_f, s0=b.src(fine))
            lv = Val(cat, fine_level, n_f + fine.cols)
        else:
            lv = b.tap_product(fm_.fine, lv, fine_level, TABLE_F2C, level, fm_.norm, residual=fine)
            lv.stats = lv.stats      # the epilogue's partial sums already include the residual
        level = fine_level
        if i == model.nr_downsamples - 1:
            lv = b.fusion(fusion[2], lv, level)                                                      # models:424
            if seq:
                stop(lv)                                                                             # models:427
    for blk in model.resnet_blocks_per_up_lvl_list[i]:                                               # models:435-437
        lv = b.block(blk, lv, level)

    # ---- SliceFastCUDALatticeModule.forward (models:465)
    sl = model.slice_fast_cuda
    clasify = _need(sl.linear_clasify, "slice parameters")
    if sl.dropout is not None and sl.dropout.prob > 0.0:
        b.uses_dropout = True
    t = None
    if sl.experiment != "slice_no_deform":
        t = lv
        for m in sl.stepdown:
            t = b.gn_relu_1x1(m, t)
        t = b.gn_relu_1x1(sl.bottleneck, t)
    scores = b.gemm(level, clasify.weight.shape[0], clasify.weight, True, b.src(lv), stats=False)
    out = b.slot(ROWS_POINTS, clasify.weight.shape[0], SLOT_OUT)
    b.keep += [clasify.bias, sl.linear_pre_deltaW.weight, sl.linear_deltaW.weight, sl.linear_deltaW.bias]
    if t is not None:
        b.fused_logsm = clasify.weight.shape[0] <= 64      # the slice head can write log_softmax(scores) as well
        b.op(OP_SLICE_DEFORM, out=out, s0=b.src(t), s1=b.src(scores), bias=_p(clasify.bias),
             p=[_p(sl.linear_pre_deltaW.weight), _p(sl.linear_deltaW.weight), _p(sl.linear_deltaW.bias)] + [None] * 5)
    else:
        b.op(OP_SLICE, out=out, s0=b.src(scores), s1=None, bias=_p(clasify.bias))
    b.out_shape = (ROWS_POINTS, clasify.weight.shape[0])
    if not stopped:
        b.stop_shape = None
    return b


def _fill_src(dst, d):
    if d is None:
        dst.slot = -1
        dst.gn_stats = -1
        return
    dst.slot, dst.table, dst.level, dst.relu = d["slot"], d["table"], d["level"], d["relu"]
    dst.pad_value, dst.gn_stats, dst.gn_groups, dst.gn_eps = d["pad_value"], d["gn_stats"], d["gn_groups"], d["gn_eps"]
    dst.gn_gamma, dst.gn_beta = d["gn_gamma"], d["gn_beta"]


class FrameProgram:
    """native per-frame forward of one LNN_SEQ instance (inference)"""

    def __init__(self, model, builder):
        self.model = model
        self.nr_coarse = model.nr_downsamples
        self.stop_shape, self.out_shape = builder.stop_shape, builder.out_shape
        self.uses_dropout = builder.uses_dropout
        self._keep = builder.keep
        self.n_ops, self.n_slots, self.n_states = len(builder.ops), len(builder.slots), builder.n_states
        self.nr_states = self.n_states
        # columns of every hidden state (what a frame-sharded receiver needs to know a state's shape before it arrives)
        self._state_cols = {state: cols for rows, cols, kind, state in builder.slots if kind == SLOT_STATE_NEW}
        self.fused_logsm = bool(getattr(builder, "fused_logsm", False))
        self.last_logsm = None
        slots = (_lib.Slot * len(builder.slots))()
        for k, (rows, cols, kind, state) in enumerate(builder.slots):
            slots[k].rows, slots[k].cols, slots[k].kind, slots[k].state = rows, cols, kind, state
        ops = (_lib.Op * len(builder.ops))()
        for k, o in enumerate(builder.ops):
            d = ops[k]
            d.kind, d.cond_state, d.cond_has = o["kind"], o["cond_state"], o["cond_has"]
            d.out, d.out_col, d.n, d.stats_out = o["out"], o["out_col"], o["n"], o["stats_out"]
            _fill_src(d.s0, o["s0"])
            _fill_src(d.s1, o["s1"])
            d.residual, d.relu, d.w_is_nk, d.w, d.bias = o["residual"], o["relu"], o["w_is_nk"], o["w"], o["bias"]
            for j in range(8):
                d.p[j] = o["p"][j]
                d.i[j] = o["i"][j]
            for j in range(4):
                d.f[j] = o["f"][j]
        h = C.c_void_p()
        _lib.check(_lib.lib().tln_program_create(C.byref(h), slots, len(builder.slots), ops, len(builder.ops),
                                                 builder.n_states, self.nr_coarse), "tln_program_create")
        self._h = h
        self._v = (C.c_int64 * (self.nr_coarse + 1))()
        self.subtract_mean = model.distribute.experiment not in NO_MEAN_EXPERIMENTS

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().tln_program_destroy(h)
            except Exception:
                pass

    def reset(self):
        _lib.check(_lib.lib().tln_program_reset(self._h), "tln_program_reset")

    def memory_bytes(self):
        """device memory of the program (tln_program_memory): arena capacity / high-water mark, K1 buffer, hidden states"""
        out = (C.c_int64 * 5)()
        _lib.check(_lib.lib().tln_program_memory(self._h, out), "tln_program_memory")
        return dict(zip(("arena", "arena_high_water", "k1_buffer", "hidden_states", "total"), (int(x) for x in out)))

    def replans(self):
        """frames whose exact coarse vertex counts exceeded the planning prediction (tln_program_replans)"""
        return int(_lib.lib().tln_program_replans(self._h))

    def apply_options(self, *lattices):
        """hands the kernel-selection options in force on this host thread (options.py; none = the library's defaults) to
        this program's handle and to the lattices of the frame: the library itself keeps no process-wide switch"""
        from . import options as O
        g = O.generation()
        if getattr(self, "_opt_gen", 0) != g:
            _lib.check(_lib.lib().tln_program_set_options(self._h, O.current_ref()), "tln_program_set_options")
            self._opt_gen = g
        for ls in lattices:
            ls.apply_options()

    def _rows(self, code, n):
        return n if code == ROWS_POINTS else (4 * n if code == ROWS_POINT_ROWS else int(self._v[code]))

    def run_frame(self, ls, positions, values, reset_hashmap, early_return, keep_early=True):
        """-> (tensor, ls): the early-return lattice values [V, C] or the class scores [N, nr_classes].
        keep_early=False: an early-return frame hands back None instead of the [V, C] values — the reference returns its
        hidden-state tensor itself there (models.py:430) and its loops drop it; here the value would be a copy out of the
        program's state buffer (23 MB per frame on the headline workload)"""
        positions = positions.contiguous().float()
        n = positions.shape[0]
        if values is None or values.numel() == 0:
            values, val_dim = None, 0
        else:
            values = values.contiguous().float()
            val_dim = values.shape[1]
        lib, s = _lib.lib(), stream_ptr()
        self.apply_options(ls)
        _lib.check(lib.tln_program_begin_frame(self._h, ls._h, positions.data_ptr(),
                                               values.data_ptr() if values is not None else None, n, val_dim,
                                               1 if reset_hashmap else 0, 1 if self.subtract_mean else 0, self._v, s),
                   "tln_program_begin_frame")
        ls._csr_key = None
        ls._bins_key = None
        ls._last_indices = None
        early = bool(early_return) and self.stop_shape is not None
        rows_code, cols = self.stop_shape if early else self.out_shape
        rows = self._rows(rows_code, n)
        out = None if (early and not keep_early) else torch.empty((rows, cols), dtype=torch.float32, device="cuda")
        self.last_logsm = None
        if not early and self.fused_logsm:
            self.last_logsm = torch.empty_like(out)
            _lib.check(lib.tln_program_set_aux_out(self._h, self.last_logsm.data_ptr()), "tln_program_set_aux_out")
        _lib.check(lib.tln_program_run(self._h, 1 if early else 0, out.data_ptr() if out is not None else None, rows, cols, s),
                   "tln_program_run")
        if out is not None:
            ls.set_values(out)
        return out, ls

    def take_logsm(self):
        """log_softmax(scores) of the frame just run, when the slice head produced it (else None)"""
        t, self.last_logsm = self.last_logsm, None
        return t

    def _start(self, ls, positions, values, reset_hashmap):
        """first half of a frame's K1 (the vertex counters start their way to the host)"""
        positions = positions.contiguous().float()
        n = positions.shape[0]
        if values is None or values.numel() == 0:
            values, val_dim = None, 0
        else:
            values = values.contiguous().float()
            val_dim = values.shape[1]
        self.apply_options(ls)
        _lib.check(_lib.lib().tln_program_begin_frame_start(self._h, ls._h, positions.data_ptr(),
                                                            values.data_ptr() if values is not None else None, n,
                                                            val_dim, 1 if reset_hashmap else 0,
                                                            1 if self.subtract_mean else 0, stream_ptr()),
                   "tln_program_begin_frame_start")
        ls._csr_key = None
        ls._bins_key = None
        ls._last_indices = None
        return n, (positions, values)                 # the inputs stay alive until the frame has been enqueued

    def _finish(self, n, early_return):
        """second half: vertex counts known, coarse levels begun -> (early, output tensor)"""
        _lib.check(_lib.lib().tln_program_begin_frame_finish(self._h, self._v, stream_ptr()),
                   "tln_program_begin_frame_finish")
        early = bool(early_return) and self.stop_shape is not None
        rows_code, cols = self.stop_shape if early else self.out_shape
        return early, torch.empty((self._rows(rows_code, n), cols), dtype=torch.float32, device="cuda")

    @staticmethod
    def run_frame_group(progs, lattices, positions, values, reset_hashmap, early_return, keep_early=True):
        """2..8 sequences in lock-step on the current stream (tln_program_run_group): -> [(tensor, ls), ...]
        (keep_early=False: None instead of the early-return values, see run_frame)"""
        n = len(progs)
        # the frames of all sequences begun by ONE native call: a single batch of K1 launches (blockIdx.y = sequence),
        # one wait for the vertex counters
        lib, s = _lib.lib(), stream_ptr()
        pos = [p_.contiguous().float() for p_ in positions]
        if values[0] is None or values[0].numel() == 0:
            vals, val_dim = None, 0
        else:
            vals = [v_.contiguous().float() for v_ in values]
            val_dim = vals[0].shape[1]
            assert all(v_.shape[1] == val_dim for v_ in vals)
        for p_, ls_ in zip(progs, lattices):
            p_.apply_options(ls_)
        hs = (C.c_void_p * n)(*[p._h for p in progs])
        lh = (C.c_void_p * n)(*[ls._h for ls in lattices])
        pp = (C.c_void_p * n)(*[x.data_ptr() for x in pos])
        vp = (C.c_void_p * n)(*[x.data_ptr() for x in vals]) if vals is not None else None
        ns = (C.c_int64 * n)(*[x.shape[0] for x in pos])
        vout = (C.c_int64 * (n * MAX_LEVELS))()
        all_early = bool(early_return) and all(p.stop_shape is not None for p in progs)
        _lib.check(lib.tln_program_begin_frame_group(hs, lh, pp, vp, ns, n, val_dim, 1 if reset_hashmap else 0,
                                                     1 if progs[0].subtract_mean else 0, 0 if all_early else 1, vout, s),
                   "tln_program_begin_frame_group")
        begun = []
        for k, (p, ls) in enumerate(zip(progs, lattices)):
            ls._csr_key = None
            ls._bins_key = None
            ls._last_indices = None
            for i in range(p.nr_coarse + 1):
                p._v[i] = vout[k * MAX_LEVELS + i]
            early_k = bool(early_return) and p.stop_shape is not None
            rows_code, cols = p.stop_shape if early_k else p.out_shape
            shape = (p._rows(rows_code, int(ns[k])), cols)
            begun.append((early_k, shape, None if (early_k and not keep_early) else
                          torch.empty(shape, dtype=torch.float32, device="cuda")))
        started = (pos, vals)                          # the inputs stay alive until the frames have been enqueued
        early, shapes, outs = begun[0][0], [x[1] for x in begun], [x[2] for x in begun]
        assert all(x[0] == early for x in begun) and all(sh[1] == shapes[0][1] for sh in shapes)
        for p, o in zip(progs, outs):
            p.last_logsm = None
            if not early and p.fused_logsm:
                p.last_logsm = torch.empty_like(o)
                _lib.check(_lib.lib().tln_program_set_aux_out(p._h, p.last_logsm.data_ptr()), "tln_program_set_aux_out")
        hs = (C.c_void_p * n)(*[p._h for p in progs])
        ptrs = (C.c_void_p * n)(*[o.data_ptr() if o is not None else None for o in outs])
        rows = (C.c_int64 * n)(*[sh[0] for sh in shapes])
        _lib.check(_lib.lib().tln_program_run_group(hs, n, 1 if early else 0, ptrs, rows, shapes[0][1], stream_ptr()),
                   "tln_program_run_group")
        del started                                    # the inputs stayed alive until the frames were enqueued
        for ls, o in zip(lattices, outs):
            if o is not None:
                ls.set_values(o)
        return list(zip(outs, lattices))

    run_frame_pair = run_frame_group

    def capture_gemms(self, enable=True):
        _lib.check(_lib.lib().tln_program_capture_gemms(self._h, 1 if enable else 0), "tln_program_capture_gemms")

    def stage_timing(self, enable=True):
        _lib.check(_lib.lib().tln_program_timing(self._h, 1 if enable else 0), "tln_program_timing")

    def stage_times_ms(self):
        """(K1 distribute, K2 pool, K8 slice) of the last frame in ms; None where the frame had no such stage"""
        ms = (C.c_float * 3)()
        _lib.check(_lib.lib().tln_program_timing_read(self._h, ms), "tln_program_timing_read")
        return [None if v < 0 else float(v) for v in ms]

    def replay_gemms(self, reps=5):
        """(ms, launches, flops, algorithmic bytes) of the last frame's gather-GEMM launches replayed `reps` times
        back to back between two HIP events on the current stream"""
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        self.apply_options()
        _lib.check(_lib.lib().tln_program_replay_gemms(self._h, reps, C.byref(ms), C.byref(n), C.byref(fl),
                                                       C.byref(by), stream_ptr()), "tln_program_replay_gemms")
        return ms.value, n.value, fl.value, by.value

    @staticmethod
    def replay_gemms_group(programs, reps=5):
        """the same for the programs of a lock-step group (models.forward_group): product i of every program through
        one tln_gather_gemm_multi call, as the group issued it; launches = products"""
        for p_ in programs:
            p_.apply_options()
        hs = (C.c_void_p * len(programs))(*[p._h for p in programs])
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().tln_program_replay_gemms_group(hs, len(programs), reps, C.byref(ms), C.byref(n), C.byref(fl),
                                                             C.byref(by), stream_ptr()), "tln_program_replay_gemms_group")
        return ms.value, n.value, fl.value, by.value

    @staticmethod
    def replay_executed(programs):
        """flops the matrix cores execute for the last frame's products of these 1..8 lock-stepped programs (one extra
        pass of the launches with the kernels' step counters on: tln_program_replay_executed)"""
        for p_ in programs:
            p_.apply_options()
        hs = (C.c_void_p * len(programs))(*[p._h for p in programs])
        fl = C.c_double()
        _lib.check(_lib.lib().tln_program_replay_executed(hs, len(programs), C.byref(fl), stream_ptr()),
                   "tln_program_replay_executed")
        return fl.value

    # ---- the frame in segments (frame-sharded multi-GPU, dist.FrameShardRunner) -----------------------------------
    def state_cols(self, sid):
        return self._state_cols[sid]

    def state_ops(self, sid):
        """(first op that reads stored state `sid`, last op that writes the new one, lattice level of the state)"""
        fr, lw, lvl = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().tln_program_state_ops(self._h, sid, C.byref(fr), C.byref(lw), C.byref(lvl)),
                   "tln_program_state_ops")
        return fr.value, lw.value, lvl.value

    def run_frame_sharded(self, ls, positions, values, reset_hashmap, early_return, recv_state=None, send_state=None,
                          expect_rows=None):
        """One frame like run_frame, but cut at the fusion slots: `recv_state(sid) -> tensor [rows, C]` is called right
        before the first op that reads hidden state `sid` (the rank of the previous frame sends it),
        `send_state(sid, tensor)` right after the last op that writes the new one.  `expect_rows[level]` = vertex count
        of every level BEFORE this frame (= the row counts of the states that will arrive)."""
        positions = positions.contiguous().float()
        n = positions.shape[0]
        if values is None or values.numel() == 0:
            values, val_dim = None, 0
        else:
            values = values.contiguous().float()
            val_dim = values.shape[1]
        lib, s = _lib.lib(), stream_ptr()
        self.apply_options(ls)
        _lib.check(lib.tln_program_begin_frame(self._h, ls._h, positions.data_ptr(),
                                               values.data_ptr() if values is not None else None, n, val_dim,
                                               1 if reset_hashmap else 0, 1 if self.subtract_mean else 0, self._v, s),
                   "tln_program_begin_frame")
        ls._csr_key = None
        ls._bins_key = None
        ls._last_indices = None
        plan = sorted((self.state_ops(sid) + (sid,)) for sid in range(self.n_states))     # by first-read op
        # a state takes part in the hand-off iff it is both read (on the receiving side) and written (on the sending
        # side) by the op list: both sides walk the SAME list of states, so that the order-matched messages pair up
        plan = [(fr, lw, lvl, sid) for fr, lw, lvl, sid in plan if fr >= 0 and lw >= 0]
        # the segments must move forward: read(k) <= write(k) < read(k+1) ... — a fusion module that interleaved the ops
        # of two states would otherwise have its state set after the ops that read it
        cuts = [x for fr, lw, lvl, sid in plan for x in (fr, lw + 1)]
        if any(b < a for a, b in zip(cuts, cuts[1:])):
            raise _lib.TlnError("frame-sharded run: the hidden states' op ranges interleave (%s)" % (plan,))
        if recv_state is not None:
            for fr, lw, lvl, sid in plan:
                if fr >= 0:
                    _lib.check(lib.tln_program_state_expect(self._h, sid, int(expect_rows[lvl]), s),
                               "tln_program_state_expect")
        early = bool(early_return) and self.stop_shape is not None
        rows_code, cols = self.stop_shape if early else self.out_shape
        rows = self._rows(rows_code, n)
        out = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
        self.last_logsm = None
        if not early and self.fused_logsm:
            self.last_logsm = torch.empty_like(out)
            _lib.check(lib.tln_program_set_aux_out(self._h, self.last_logsm.data_ptr()), "tln_program_set_aux_out")
        _lib.check(lib.tln_program_run_begin(self._h, 1 if early else 0, out.data_ptr(), rows, cols, s),
                   "tln_program_run_begin")
        for fr, lw, lvl, sid in plan:
            if recv_state is not None and fr >= 0:
                _lib.check(lib.tln_program_run_until(self._h, fr, s), "tln_program_run_until")
                h = recv_state(sid)
                if h is not None and h.numel():
                    h = h.contiguous().float()
                    _lib.check(lib.tln_program_state_set(self._h, sid, h.data_ptr(), h.shape[0], s),
                               "tln_program_state_set")
                    self._keep_recv = h        # alive until the copy on the stream has been enqueued
            if send_state is not None and lw >= 0:
                _lib.check(lib.tln_program_run_until(self._h, lw + 1, s), "tln_program_run_until")
                r, c, w = C.c_int64(), C.c_int(), C.c_int()
                _lib.check(lib.tln_program_state_new_info(self._h, sid, C.byref(r), C.byref(c), C.byref(w)),
                           "tln_program_state_new_info")
                if w.value:
                    t = torch.empty((r.value, c.value), dtype=torch.float32, device="cuda")
                    _lib.check(lib.tln_program_state_get_new(self._h, sid, t.data_ptr(), s), "tln_program_state_get_new")
                else:
                    t = torch.zeros((0,), dtype=torch.float32, device="cuda")
                send_state(sid, t)
        _lib.check(lib.tln_program_run_end(self._h, s), "tln_program_run_end")
        ls.set_values(out)
        return out, ls

    def state(self, sid):
        """copy of hidden state `sid` (None if it does not exist yet)"""
        rows, cols, has = C.c_int64(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().tln_program_state_info(self._h, sid, C.byref(rows), C.byref(cols), C.byref(has)),
                   "tln_program_state_info")
        if not has.value:
            return None
        t = torch.empty((rows.value, cols.value), dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().tln_program_state_get(self._h, sid, t.data_ptr(), stream_ptr()), "tln_program_state_get")
        return t


def params_key(model, refresh=False):
    """identity + in-place version of every parameter: a program is rebuilt when any of them was re-allocated or
    written (optimizer step, load_state_dict, .to()) since it was compiled — AFlow's alpha / beta are baked in by
    value.  The list of Parameter objects is walked once per compile (a module-tree walk per sequence costs more host
    time than the sequence's native calls); assigning a NEW Parameter object to a module attribute afterwards needs
    `model._program = None`."""
    plist = None if refresh else getattr(model, "_program_params", None)
    if plist is None:
        plist = list(model.parameters())
        object.__setattr__(model, "_program_params", plist)
    return tuple([(p.data_ptr(), p._version) for p in plist])


def compile_model(model):
    """FrameProgram for `model`, or None when the configuration / state needs the operator-level route"""
    try:
        b = _walk_model(model)
    except NotReady:
        return None
    except Unsupported:
        object.__setattr__(model, "_program_unsupported", True)     # the configuration will not change: stop trying
        return None
    return FrameProgram(model, b)
