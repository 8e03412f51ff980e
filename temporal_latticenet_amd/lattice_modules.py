"""The operator API the reference imports from `latticenet_py.lattice.lattice_modules` / `lattice_funcs`
(un-vendored; names and call signatures taken from the call sites in seq_lattice/models.py:62, 175-234, 298,
353, 398, 465 and seq_lattice/lattice_modules.py:75-76, 100, 301-304, 436-440, 573).

Every module lazily creates its parameters on the first forward, like the reference's own modules do
(lattice_modules.py:288-295, 410-440), so a checkpoint is loaded after the first forward (train_ln.py:193-209).
The arithmetic is done by the gfx950 kernels behind include/tln.h; the semantics of each module are fixed in
DESIGN.md ("Lattice specification").
"""
import math

import torch

from . import autograd as AG
from . import ops
from .lattice import Lattice

__all__ = [
    "DistributeLatticeModule", "ConvLatticeModule", "CoarsenLatticeModule", "FinefyLatticeModule",
    "GroupNormLatticeModule", "Gn", "GnRelu1x1", "Conv1x1", "GnReluConv", "GnReluCoarsen", "GnReluFinefy",
    "ResnetBlock", "BottleneckBlock", "SliceLatticeModule", "SplatLatticeModule", "SliceFastCUDALatticeModule",
    "DropoutLattice", "Im2RowLattice", "Im2RowIndicesLattice", "gn_groups",
]

NO_MEAN_EXPERIMENTS = ("pointnet_no_local_mean", "pointnet_no_elevate_no_local_mean", "splat")


def gn_groups(c):
    return 32 if c % 32 == 0 else max(c // 2, 1)


def _kaiming_uniform_fan_out_(weight, fan_out):
    """the reference's own initialiser for im2row conv weights (lattice_modules.py:264-272)"""
    gain = torch.nn.init.calculate_gain("relu", 1)
    bound = math.sqrt(3.0) * gain / math.sqrt(fan_out)
    with torch.no_grad():
        weight.uniform_(-bound, bound)


# --------------------------------------------------------------------------------------------
# structure
# --------------------------------------------------------------------------------------------
class DistributeLatticeModule(torch.nn.Module):
    """models.py:62, 298: (ls, positions, values, reset_hashmap) -> (ls, distributed, indices, weights)"""

    def __init__(self, experiment="none"):
        super().__init__()
        self.experiment = experiment

    def forward(self, lattice, positions, values, reset_hashmap=True):
        with torch.no_grad():
            d, i, w = lattice.distribute(positions, values, reset_hashmap,
                                         subtract_mean=self.experiment not in NO_MEAN_EXPERIMENTS)
        return lattice, d, i, w


class Im2RowLattice:
    """Im2RowLattice.apply(lv, ls, filter_extent, dilation, nr_filters) -> [V, 9*C] (lm:301)."""

    @staticmethod
    def apply(lattice_values, lattice, filter_extent=9, dilation=1, nr_filters=None):
        return ops.im2row(lattice_values, lattice.neighbour_table_ptr(), lattice.nr_lattice_vertices())


class Im2RowIndicesLattice:
    """Im2RowIndicesLattice.apply(...) -> [V, 9*C] ints whose [:, ::C] is the [V,9] neighbour table (lm:304, 318)."""

    @staticmethod
    def apply(lattice_values, lattice, filter_extent=9, dilation=1, nr_filters=None):
        c = lattice_values.shape[1] if nr_filters is None else nr_filters
        return lattice.neighbour_table().repeat_interleave(c, dim=1)


# --------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------
class GroupNormLatticeModule(torch.nn.Module):
    """GroupNorm whose statistics run over all vertices x channels of a group (layout [1,C,V])."""

    def __init__(self, nr_params=None, affine=True):
        super().__init__()
        self.norm = None
        self.affine = affine
        if nr_params is not None:
            self._make(nr_params)

    def _make(self, c):
        self.norm = torch.nn.GroupNorm(gn_groups(c), c, affine=self.affine).to("cuda")

    def ensure(self, lv):
        if self.norm is None:
            self._make(lv.shape[1])
        return self.norm

    def stats(self, lv):
        self.ensure(lv)
        return ops.groupnorm_stats(lv, self.norm.num_groups, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, lv, ls, relu=False):
        if AG.grad_mode():                       # training path: torch's differentiable GroupNorm
            out = AG.group_norm_relu(lv, self.ensure(lv), relu)
            ls.set_values(out)
            return out, ls
        scale, shift = self.stats(lv)
        out = ops.affine_act(lv, scale, shift, relu)
        ls.set_values(out)
        return out, ls


class Gn(GroupNormLatticeModule):
    """Gn() of the reference (lm:75, 100)."""

    def __init__(self):
        super().__init__(None, True)


# --------------------------------------------------------------------------------------------
# dense products
# --------------------------------------------------------------------------------------------
class Conv1x1(torch.nn.Module):
    """Conv1x1(out_channels, bias) (lm:76): a lazily created Linear applied per vertex."""

    def __init__(self, out_channels, bias=True):
        super().__init__()
        self.out_channels = out_channels
        self.use_bias = bias
        self.linear = None

    def _make(self, cin):
        self.linear = torch.nn.Linear(cin, self.out_channels, bias=self.use_bias).to("cuda")
        with torch.no_grad():
            torch.nn.init.kaiming_normal_(self.linear.weight, mode="fan_in", nonlinearity="relu")
            if self.use_bias:
                self.linear.bias.zero_()

    def forward(self, lv, ls, prologue=None, residual=None):
        if self.linear is None:
            self._make(lv.shape[1])
        if AG.grad_mode():                       # training path (the caller has applied GN/ReLU already)
            assert prologue is None
            out = AG.gather_gemm(lv.shape[0], lv, None, self.linear.weight, self.linear.bias, residual, w_is_nk=True)
            ls.set_values(out)
            return out, ls
        gn = None
        if prologue is not None and prologue[0] == "gn":      # GroupNorm+ReLU of `lv` inside the same host call
            gn, prologue = (lv, prologue[1].ensure(lv), True), None
        scale, shift, relu = prologue if prologue is not None else (None, None, False)
        src = ops.gemm_src(lv, scale=scale, shift=shift, relu=relu)
        out = ops.gather_gemm(lv.shape[0], self.linear.weight, src, w_is_nk=True, bias=self.linear.bias,
                              residual=residual, stats=True, gn=gn)
        ls.set_values(out)
        return out, ls


class _TapConv(torch.nn.Module):
    """shared body of the 9-tap products: weight [9*C_in, nr_filters] (layout of lm:291)"""

    def __init__(self, nr_filters, bias):
        super().__init__()
        self.nr_filters = nr_filters
        self.use_bias = bias
        self.weight = None
        self.bias = None

    def _make(self, cin):
        self.weight = torch.nn.Parameter(torch.empty(9 * cin, self.nr_filters, device="cuda"))
        _kaiming_uniform_fan_out_(self.weight, self.nr_filters)
        if self.use_bias:
            bound = 1.0 / math.sqrt(self.nr_filters)
            self.bias = torch.nn.Parameter(torch.empty(self.nr_filters, device="cuda").uniform_(-bound, bound))

    def _product(self, rows, lv, table_ptr, prologue, residual, table_tensor=None, symmetric=False, src_lattice=None):
        if self.weight is None:
            self._make(lv.shape[1])
        if AG.grad_mode():                       # training path: autograd wrapper around the same kernel
            assert prologue is None
            return AG.gather_gemm(rows, lv, table_tensor(), self.weight, self.bias, residual, symmetric=symmetric,
                                  src_lattice=src_lattice)
        gn = None
        if prologue is not None and prologue[0] == "gn":      # GroupNorm+ReLU of `lv` inside the same host call
            gn, prologue = (lv, prologue[1].ensure(lv), True), None
        scale, shift, relu = prologue if prologue is not None else (None, None, False)
        src = ops.gemm_src(lv, table_ptr, 9, scale=scale, shift=shift, relu=relu)
        # stats=True: the product also emits the GroupNorm partial sums of its output for whatever Gn comes next
        return ops.gather_gemm(rows, self.weight, src, bias=self.bias, residual=residual, stats=True, gn=gn)


class ConvLatticeModule(_TapConv):
    """ConvLatticeModule(nr_filters, neighbourhood_size, dilation, bias) (lm:440, 573)."""

    def __init__(self, nr_filters, neighbourhood_size=1, dilation=1, bias=True):
        super().__init__(nr_filters, bias)
        assert neighbourhood_size == 1 and dilation == 1, "only the one-hop, dilation-1 filter is supported"

    def forward(self, lv, ls, prologue=None, residual=None):
        out = self._product(ls.nr_lattice_vertices(), lv, ls.neighbour_table_ptr(), prologue, residual,
                            ls.neighbour_table, symmetric=True)
        ls.set_values(out)
        return out, ls


class CoarsenLatticeModule(_TapConv):
    def __init__(self, nr_filters, bias=False):
        super().__init__(nr_filters, bias)

    def forward(self, lv, ls, prologue=None):
        coarse = ls.coarsen()
        out = self._product(coarse.nr_lattice_vertices(), lv, coarse.coarse_to_fine_table_ptr(), prologue, None,
                            coarse.coarse_to_fine_table, src_lattice=ls)
        coarse.set_values(out)
        return out, coarse


class FinefyLatticeModule(_TapConv):
    def __init__(self, nr_filters, bias=False):
        super().__init__(nr_filters, bias)

    def forward(self, lv_coarse, ls_coarse, ls_fine, prologue=None):
        nf = ls_fine.nr_lattice_vertices()
        out = self._product(nf, lv_coarse, ls_coarse.fine_to_coarse_table_ptr(), prologue, None,
                            lambda: ls_coarse.fine_to_coarse_table(nf), src_lattice=ls_coarse)
        ls_fine.set_values(out)
        return out, ls_fine


class DropoutLattice(torch.nn.Module):
    def __init__(self, prob):
        super().__init__()
        self.prob = prob

    def forward(self, lv, ls):
        if self.training and self.prob > 0.0:
            lv = torch.nn.functional.dropout(lv, self.prob, True)
            ls.set_values(lv)
        return lv, ls


# --------------------------------------------------------------------------------------------
# GroupNorm -> ReLU -> product, with the normalisation folded into the product's operand staging
# --------------------------------------------------------------------------------------------
class GnRelu1x1(torch.nn.Module):
    """GnRelu1x1(out_channels, bias) (lm:436-437)"""

    def __init__(self, out_channels, bias):
        super().__init__()
        self.norm = Gn()
        self.linear = Conv1x1(out_channels, bias)

    def forward(self, lv, ls, residual=None):
        if AG.grad_mode():
            return self.linear(AG.group_norm_relu(lv, self.norm.ensure(lv)), ls, None, residual)
        return self.linear(lv, ls, ("gn", self.norm), residual)


class GnReluConv(torch.nn.Module):
    def __init__(self, nr_filters, dilation=1, bias=False, with_dropout=False):
        super().__init__()
        self.norm = Gn()
        self.conv = ConvLatticeModule(nr_filters, 1, dilation, bias)
        self.drop = DropoutLattice(0.2) if with_dropout else None

    def forward(self, lv, ls, residual=None):
        if self.drop is not None and self.training:
            lv, ls = self.norm(lv, ls, relu=True)
            lv, ls = self.drop(lv, ls)
            return self.conv(lv, ls, None, residual)
        if AG.grad_mode():
            return self.conv(AG.group_norm_relu(lv, self.norm.ensure(lv)), ls, None, residual)
        return self.conv(lv, ls, ("gn", self.norm), residual)


class GnReluCoarsen(torch.nn.Module):
    """GnReluCoarsen(nr_filters) (models.py:182, 353): (lv, ls) -> (lv_coarse, ls_coarse)"""

    def __init__(self, nr_filters):
        super().__init__()
        self.norm = Gn()
        self.coarse = CoarsenLatticeModule(nr_filters, bias=False)

    def forward(self, lv, ls):
        if AG.grad_mode():
            return self.coarse(AG.group_norm_relu(lv, self.norm.ensure(lv)), ls, None)
        return self.coarse(lv, ls, ("gn", self.norm))


class GnReluFinefy(torch.nn.Module):
    """GnReluFinefy(nr_filters) (models.py:214, 398): (lv_coarse, ls_coarse, ls_fine) -> (lv_fine, ls_fine)"""

    def __init__(self, nr_filters):
        super().__init__()
        self.norm = Gn()
        self.fine = FinefyLatticeModule(nr_filters, bias=False)

    def forward(self, lv_coarse, ls_coarse, ls_fine):
        if AG.grad_mode():
            return self.fine(AG.group_norm_relu(lv_coarse, self.norm.ensure(lv_coarse)), ls_coarse, ls_fine, None)
        return self.fine(lv_coarse, ls_coarse, ls_fine, ("gn", self.norm))


class ResnetBlock(torch.nn.Module):
    """ResnetBlock(nr_filters, dilations, biases, with_dropout) (models.py:175, 227)"""

    def __init__(self, nr_filters, dilations, biases, with_dropout):
        super().__init__()
        self.conv1 = GnReluConv(nr_filters, dilations[0], biases[0], with_dropout=False)
        self.conv2 = GnReluConv(nr_filters, dilations[1], biases[1], with_dropout=with_dropout)

    def forward(self, lv, ls):
        identity = lv
        lv, ls = self.conv1(lv, ls)
        lv, ls = self.conv2(lv, ls, residual=identity)
        return lv, ls


class BottleneckBlock(torch.nn.Module):
    """BottleneckBlock(out_channels, biases) (models.py:178, 193, 230): 1x1 C->C/4, conv, 1x1 C/4->C, + identity"""

    def __init__(self, out_channels, biases):
        super().__init__()
        self.downsample = 4
        self.contract = GnRelu1x1(int(out_channels / self.downsample), biases[0])
        self.conv = GnReluConv(int(out_channels / self.downsample), 1, biases[1], with_dropout=False)
        self.expand = GnRelu1x1(out_channels, biases[2])

    def forward(self, lv, ls):
        identity = lv
        lv, ls = self.contract(lv, ls)
        lv, ls = self.conv(lv, ls)
        lv, ls = self.expand(lv, ls, residual=identity)
        return lv, ls


# --------------------------------------------------------------------------------------------
# splat / slice
# --------------------------------------------------------------------------------------------
class SplatLatticeModule(torch.nn.Module):
    """plain splat: lv[v] = sum_rows w * [values, 1] (homogeneous coordinate last)"""

    def forward(self, lattice, positions, values, reset_hashmap=True):
        with torch.no_grad():
            _, indices, weights = lattice.distribute(positions, values, reset_hashmap, subtract_mean=False)
        lv = ops.splat(lattice, values, indices, weights)
        lattice.set_values(lv)
        return lv, lattice, indices, weights


class SliceLatticeModule(torch.nn.Module):
    """plain slice: out[p] = sum_r w_r * lv[idx_r]"""

    def forward(self, lv, ls, positions, indices, weights):
        ls.set_values(lv)
        return ops.slice_blend(lv, indices, weights)


class SliceFastCUDALatticeModule(torch.nn.Module):
    """SliceFastCUDALatticeModule(nr_classes, dropout_prob, experiment) (models.py:232, 465).

    DeformSlice of LatticeNet: a per-vertex bottleneck (two GnRelu1x1 step-downs C -> C -> C/2, then 8 channels)
    is gathered per point over the d+1 simplex vertices ([w*b, w] each), a two-layer head predicts offsets dw to
    the barycentric weights, and the point's class scores are sum_r (w_r + dw_r) * (lv[idx_r] @ Wc^T) + bc.
    The classifier runs per VERTEX on the matrix cores before the blend (V << N), which is algebraically the
    fused slice-classify of the reference."""

    BOTTLENECK = 8

    def __init__(self, nr_classes, dropout_prob=0.0, experiment="none"):
        super().__init__()
        self.nr_classes = nr_classes
        self.experiment = experiment
        self.dropout = DropoutLattice(dropout_prob) if dropout_prob > 0.0 else None
        self.stepdown = torch.nn.ModuleList([])
        self.bottleneck = GnRelu1x1(self.BOTTLENECK, False)
        self.linear_pre_deltaW = None
        self.linear_deltaW = None
        self.linear_clasify = None

    def _make(self, val_dim):
        for i in range(2):
            nr = int(val_dim / (2 ** i))
            if nr < self.BOTTLENECK:
                raise ValueError("slice step-down would go below the bottleneck size")
            self.stepdown.append(GnRelu1x1(nr, False))
        g = 4 * (self.BOTTLENECK + 1)
        self.linear_pre_deltaW = torch.nn.Linear(g, g, bias=False).to("cuda")
        self.linear_deltaW = torch.nn.Linear(g, 4, bias=True).to("cuda")
        self.linear_clasify = torch.nn.Linear(val_dim, self.nr_classes, bias=True).to("cuda")
        with torch.no_grad():
            torch.nn.init.kaiming_uniform_(self.linear_pre_deltaW.weight, mode="fan_in", nonlinearity="relu")
            self.linear_deltaW.weight.mul_(0.1)
            self.linear_deltaW.bias.zero_()

    def forward(self, lv, ls, positions, indices, weights):
        if self.linear_clasify is None:
            self._make(lv.shape[1])
        ls.set_values(lv)
        if self.dropout is not None:
            lv, ls = self.dropout(lv, ls)
        if AG.grad_mode():                       # training path: same arithmetic in differentiable torch form
            delta = None
            if self.experiment != "slice_no_deform":
                b = lv
                for m in self.stepdown:
                    b, _ = m(b, ls)
                b, _ = self.bottleneck(b, ls)
                g = AG.slice_gather(b, indices, weights, lattice=ls)
                hdn = torch.relu(torch.nn.functional.linear(g, self.linear_pre_deltaW.weight))
                delta = torch.nn.functional.linear(hdn, self.linear_deltaW.weight, self.linear_deltaW.bias)
            feat = AG.slice_blend(lv, indices, weights, delta, lattice=ls)
            ls.set_values(lv)
            return torch.nn.functional.linear(feat, self.linear_clasify.weight, self.linear_clasify.bias)
        b = None
        if self.experiment != "slice_no_deform":
            b = lv
            for m in self.stepdown:
                b, _ = m(b, ls)
            b, _ = self.bottleneck(b, ls)
        scores = ops.gather_gemm(lv.shape[0], self.linear_clasify.weight, ops.gemm_src(lv), w_is_nk=True)
        if b is not None:   # gather -> Linear(36,36)+ReLU -> Linear(36,4) -> blend, per point, in one kernel
            out = ops.slice_deform(b, scores, indices, weights, self.linear_pre_deltaW.weight,
                                   self.linear_deltaW.weight, self.linear_deltaW.bias, self.linear_clasify.bias)
        else:
            out = ops.slice_blend(scores, indices, weights, None, self.linear_clasify.bias)
        ls.set_values(lv)
        return out
