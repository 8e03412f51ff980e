"""Host-side mirror of the reference's seq_lattice/lattice_modules.py (same class names, constructor
signatures, attribute / parameter names and sequence-state behaviour), with the arithmetic routed to the
gfx950 kernels behind include/tln.h.  Reference lines are cited per class.

Parameter names match the reference so that its checkpoints load: `GRU.weight_ih`, `hidden_linear.weight`,
`lstm.*`, `AFLOW.{alpha,beta,weight,bias}`, `linear.*`, `layers.{0,1,2}.*`, `fusion_module.*`.
"""
import math

import torch

from . import autograd as AG
from . import ops
from .lattice_modules import (Conv1x1, ConvLatticeModule, Gn, GnRelu1x1, Im2RowIndicesLattice, Im2RowLattice)

__all__ = ["LSTMModule", "GRUModule", "CrossframeGlobalAttentionModule", "TemporalMaxPoolModule",
           "TemporalLinearModule", "CrossframeLocalInterpolationModule", "CustomKernelConvLatticeIm2RowModule",
           "PointNetSeqModule"]


def _keep(t):
    """the hidden state kept for the next frame.  The reference clones (lm:30, 56, 63, ...); here nothing mutates a
    fusion output in place (PointNetSeqModule copies before zeroing row 0), so inference can alias.  The training
    path keeps the clone: autograd needs the version counters untouched."""
    return t.clone() if AG.grad_mode() else t


def _linear(mod, x, src_rows=None, pad_value=0.0, relu=False, rows=None):
    """torch.nn.Linear `mod` applied per vertex on the matrix cores; rows beyond src_rows read as pad_value"""
    rows = x.shape[0] if rows is None else rows
    if AG.grad_mode():                           # training path
        y = torch.nn.functional.linear(AG.pad_rows(x, rows, pad_value), mod.weight, mod.bias)
        return torch.relu(y) if relu else y
    return ops.gather_gemm(rows, mod.weight, ops.gemm_src(x, src_rows=src_rows, pad_value=pad_value), w_is_nk=True,
                           bias=mod.bias, relu=relu)


class LSTMModule(torch.nn.Module):
    """reference lm:17-40"""

    def __init__(self, nr_output_channels):
        super().__init__()
        self.lstm = torch.nn.LSTMCell(input_size=nr_output_channels, hidden_size=nr_output_channels, bias=True)
        self.hidden_linear = torch.nn.Linear(nr_output_channels, nr_output_channels)
        self.h_lv = None

    def reset_sequence(self):
        self.h_lv = None

    def forward(self, lv, ls):
        if self.h_lv is None:
            self.h_lv = _keep(lv)
        else:
            self.h_lv = _linear(self.hidden_linear, self.h_lv)                       # lm:32
            V, C = lv.shape
            Vh = self.h_lv.shape[0]
            if AG.grad_mode():
                Fn = torch.nn.functional
                gates = Fn.linear(lv, self.lstm.weight_ih, self.lstm.bias_ih) + \
                    Fn.linear(AG.pad_rows(self.h_lv, V), self.lstm.weight_hh, self.lstm.bias_hh)
                i, f, g, o = gates.chunk(4, 1)
                lv = torch.sigmoid(o) * torch.tanh(torch.sigmoid(i) * torch.tanh(g))
                self.h_lv = _keep(lv)
                ls.set_values(lv)
                return lv, ls
            # gates = lv W_ih^T + b_ih + pad(h) W_hh^T + b_hh  (cell state is zero, lm:36)
            gi = ops.gather_gemm(V, self.lstm.weight_ih, ops.gemm_src(lv), w_is_nk=True, bias=self.lstm.bias_ih)
            gates = ops.gather_gemm(V, self.lstm.weight_hh, ops.gemm_src(self.h_lv, src_rows=Vh, pad_value=0.0),
                                    w_is_nk=True, bias=self.lstm.bias_hh, residual=gi)
            lv = ops.lstm_gates(gates, C)                 # sig(o) * tanh(sig(i) * tanh(g)): f * c0 with c0 = 0 (lm:36-38)
            self.h_lv = _keep(lv)
            ls.set_values(lv)
        return lv, ls


class GRUModule(torch.nn.Module):
    """reference lm:42-66"""

    def __init__(self, nr_output_channels):
        super().__init__()
        self.GRU = torch.nn.GRUCell(input_size=nr_output_channels, hidden_size=nr_output_channels, bias=True)
        self.hidden_linear = torch.nn.Linear(nr_output_channels, nr_output_channels)
        self.h_lv = None

    def reset_sequence(self):
        self.h_lv = None

    def forward(self, lv, ls):
        if self.h_lv is None:                                                        # lm:54-56
            new_lv = lv if not AG.grad_mode() else lv.clone()
            self.h_lv = _keep(lv)
        else:
            self.h_lv = _linear(self.hidden_linear, self.h_lv)                       # lm:58
            # zero padding of h to lv.shape[0] rows (lm:59-60) happens inside the kernel
            if AG.grad_mode():
                new_lv = AG.gru_cell(lv, AG.pad_rows(self.h_lv, lv.shape[0]), self.GRU)
            else:
                new_lv = ops.gru_cell(lv, self.h_lv, self.GRU.weight_ih, self.GRU.weight_hh, self.GRU.bias_ih,
                                      self.GRU.bias_hh)                              # lm:62
            self.h_lv = _keep(new_lv)
            ls.set_values(new_lv)
        return new_lv, ls


class CrossframeGlobalAttentionModule(torch.nn.Module):
    """reference lm:70-116"""

    def __init__(self, nr_output_channels):
        super().__init__()
        self.relu = torch.nn.ReLU(inplace=False)
        self.sigmoid = torch.nn.Sigmoid()
        self.groupnorm = Gn()
        self.conv = Conv1x1(out_channels=nr_output_channels, bias=False)
        self.hidden_linear = torch.nn.Linear(nr_output_channels, nr_output_channels)
        self.h_lv = None

    def reset_sequence(self):
        self.h_lv = None

    def forward(self, lv, ls):
        if self.h_lv is None:
            self.h_lv = _keep(lv)
        else:
            V = lv.shape[0]
            Vh = self.h_lv.shape[0]
            self.h_lv = _linear(self.hidden_linear, self.h_lv)                       # lm:89
            if AG.grad_mode():
                h_lv = torch.nn.functional.pad(self.h_lv, (0, 0, 0, V - Vh), value=0.0)  # lm:90-91
                h_lv, _ = self.conv(h_lv, ls)                                        # lm:95
                h_lv = self.relu(h_lv)                                               # lm:98
                h_lv, _ = self.groupnorm(h_lv, ls)                                   # lm:100
                h_lv, _ = self.conv(h_lv, ls)                                        # lm:102
                h_lv = h_lv * (1.0 / (h_lv.shape[0] + h_lv.shape[1]))                # lm:104
                h_lv = self.sigmoid(h_lv)                                            # lm:106
                if V > Vh:                                                           # lm:109-110
                    h_lv = torch.cat([h_lv[:Vh], torch.ones_like(h_lv[Vh:])], 0)
                lv = h_lv * lv                                                       # lm:112
            else:
                # the same chain on the library's kernels: conv1x1 of the zero-padded state with the ReLU in the
                # epilogue (lm:90-98), GroupNorm folded into the second conv1x1 (lm:100-102), then scale, sigmoid,
                # ones for the rows born in this frame and the gate in one element-wise pass (lm:104-112)
                if self.conv.linear is None:
                    self.conv._make(self.h_lv.shape[1])
                w = self.conv.linear.weight
                a = ops.gather_gemm(V, w, ops.gemm_src(self.h_lv, src_rows=Vh, pad_value=0.0), w_is_nk=True, relu=True,
                                    stats=True)
                a = ops.gather_gemm(V, w, ops.gemm_src(a), w_is_nk=True, gn=(a, self.groupnorm.ensure(a), False))
                lv = ops.cga_gate(a, lv, Vh, 1.0 / (V + a.shape[1]))
            self.h_lv = _keep(lv)
            ls.set_values(lv)
        return lv, ls


class TemporalMaxPoolModule(torch.nn.Module):
    """reference lm:119-145"""

    def __init__(self):
        super().__init__()
        self.h_lv = None

    def reset_sequence(self):
        self.h_lv = None

    def forward(self, lv, ls):
        alpha = 0.0
        if self.h_lv is None:
            self.h_lv = _keep(lv)
        else:
            if AG.grad_mode() or alpha != 0.0:
                pad = lv.shape[0] - self.h_lv.shape[0]
                h_lv = torch.nn.functional.pad(self.h_lv, (0, 0, 0, pad), value=-9999.0)  # lm:138-139
                lv = torch.maximum(h_lv, lv)                                         # lm:141
                self.h_lv = _keep(lv) if alpha == 0.0 else alpha * h_lv + (1 - alpha) * lv   # lm:142
            else:
                lv = ops.temporal_max(lv, self.h_lv, -9999.0)                        # lm:138-141 in one kernel
                self.h_lv = _keep(lv)
        ls.set_values(lv)
        return lv, ls


class TemporalLinearModule(torch.nn.Module):
    """reference lm:149-185"""

    def __init__(self, nr_output_channels):
        super().__init__()
        self.nr_output_channels = nr_output_channels
        self.relu = torch.nn.ReLU(inplace=False)
        self.linear = torch.nn.Linear(self.nr_output_channels * 2, self.nr_output_channels)
        self.hidden_linear = torch.nn.Linear(nr_output_channels, nr_output_channels)
        self.h_lv = None

    def reset_sequence(self):
        self.h_lv = None

    def forward(self, lv, ls):
        if self.h_lv is None:
            if lv.shape[1] != self.nr_output_channels:
                print("The second dimension of lv and the MLP do not match! Lv is ", lv.shape[1],
                      "output channels is ", self.nr_output_channels)
                exit(1)
            self.h_lv = _keep(lv)
        else:
            self.h_lv = _linear(self.hidden_linear, self.h_lv)                       # lm:172
            V, Vh = lv.shape[0], self.h_lv.shape[0]
            # cat([pad(h), lv]) @ W^T + b, ReLU (lm:174-179); alpha = 0 removes the h term of lm:181
            if AG.grad_mode():
                lv = torch.relu(torch.nn.functional.linear(torch.cat([AG.pad_rows(self.h_lv, V), lv], 1),
                                                           self.linear.weight, self.linear.bias))
            else:
              lv = ops.gather_gemm(V, self.linear.weight, ops.gemm_src(self.h_lv, src_rows=Vh, pad_value=0.0),
                                 ops.gemm_src(lv), w_is_nk=True, bias=self.linear.bias, relu=True, stats=True)
            self.h_lv = _keep(lv)
        ls.set_values(lv)
        return lv, ls


class CustomKernelConvLatticeIm2RowModule(torch.nn.Module):
    """AFlow core, reference lm:238-339.  `weight` is created and initialised but never used, exactly like
    the reference (lm:291-295) so that checkpoints keep their key set."""

    def __init__(self, nr_filters, neighbourhood_size=1, dilation=1, bias=True, use_center=True,
                 train_alpha_beta=True):
        super().__init__()
        self.first_time = True
        self.weight = None
        self.bias = None
        self.neighbourhood_size = neighbourhood_size
        self.nr_filters = nr_filters
        self.dilation = dilation
        self.use_bias = bias
        self.use_center = use_center
        if train_alpha_beta:
            print("AFLOW: Training alpha and beta values")
            self.alpha = torch.nn.Parameter(data=torch.tensor(0.1), requires_grad=True)
            self.beta = torch.nn.Parameter(data=torch.tensor(0.1), requires_grad=True)
        else:
            print("AFLOW: alpha and beta are set to 0.1 (constant)")
            self.alpha = torch.tensor(0.1)
            self.beta = torch.tensor(0.1)
        self.weights = None
        self.counter = 0

    def reset_parameters(self, filter_extent):                                       # lm:264-278
        fan = torch.nn.init._calculate_correct_fan(self.weight, "fan_out")
        gain = torch.nn.init.calculate_gain("relu", 1)
        std = gain / math.sqrt(fan)
        bound = math.sqrt(3.0) * std
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            fan_in, fan_out = torch.nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_out)
            torch.nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, lattice_values, hidden_state, lattice_structure):
        lattice_structure.set_values(lattice_values)                                 # lm:284
        filter_extent = lattice_structure.get_filter_extent(self.neighbourhood_size)
        if self.first_time:                                                          # lm:288-295
            self.first_time = False
            val_dim = lattice_structure.val_dim()
            self.weight = torch.nn.Parameter(torch.empty(filter_extent * val_dim, self.nr_filters).to("cuda"))
            if self.use_bias:
                self.bias = torch.nn.Parameter(torch.empty(self.nr_filters).to("cuda"))
            with torch.no_grad():
                self.reset_parameters(filter_extent)
        if AG.grad_mode():                       # training path: lm:298-334 in differentiable torch form
            V, Cn = lattice_values.shape
            table = lattice_structure.neighbour_table()
            hp = AG.pad_rows(hidden_state, V, -999999.0)
            nbrs = AG.im2row(hp, table, symmetric=True).reshape(V, 9, Cn)      # the level's own table: taps pair up
            valid = (table != -1).float()
            d = torch.cdist(nbrs, lattice_values.unsqueeze(1), p=2.0).squeeze(2) * valid
            if not self.use_center:
                d = torch.cat([d[:, :-1], d[:, -1:] * 0.0], 1)
            d = d * 1 / (torch.sum(d, dim=1, keepdim=True).detach())
            alpha_t = torch.ones_like(d) * self.alpha
            weights = (alpha_t - torch.min(d, alpha_t)) * self.beta * valid
            if not self.use_center:
                weights = torch.cat([weights[:, :-1], weights[:, -1:] * 0.0], 1)
            out = (nbrs * weights.unsqueeze(2)).sum(1)
            if self.use_bias:
                out = out + self.bias
            lattice_structure.set_values(lattice_values)
            return out, weights, table
        # `hidden_state` is the -999999-padded h^(t-1) (lm:215); the kernel applies the same padding to rows
        # beyond hidden_state's own length, so an unpadded tensor is accepted as well.
        out, weights, nbr_idx = ops.aflow(lattice_values, hidden_state, lattice_structure.neighbour_table_ptr(),
                                          float(self.alpha), float(self.beta), self.bias, -999999.0,
                                          self.use_center)                           # lm:298-334
        lattice_structure.set_values(lattice_values)                                 # lm:338
        return out, weights, nbr_idx


class CrossframeLocalInterpolationModule(torch.nn.Module):
    """AFlow wrapper, reference lm:188-235"""

    def __init__(self, nr_output_channels, train_alpha_beta=True, use_center=True):
        super().__init__()
        self.h_lv = None
        self.nr_output_channels = nr_output_channels
        self.AFLOW = CustomKernelConvLatticeIm2RowModule(nr_filters=nr_output_channels,
                                                         train_alpha_beta=train_alpha_beta, use_center=use_center)
        self.relu = torch.nn.ReLU(inplace=False)
        self.linear = torch.nn.Linear(self.nr_output_channels * 2, self.nr_output_channels)
        self.h_lv_vis, self.weights_vis, self.lattice_neighbors_previous = None, None, None

    def reset_sequence(self):
        self.h_lv = None
        self.h_lv_vis, self.weights_vis, self.lattice_neighbors_previous = None, None, None

    def return_for_vis(self):
        return self.h_lv_vis, self.weights_vis, self.lattice_neighbors_previous

    def forward(self, lv, ls):
        if self.h_lv is None:
            self.h_lv = _keep(lv)
        else:
            # h is padded with -999999 inside the kernel (lm:213-215); the padded copy is only kept for the
            # visualiser hooks (lm:219)
            aflow_vec, weights, nbr_prev = self.AFLOW(lv, self.h_lv, ls)             # lm:218
            self.h_lv_vis, self.weights_vis, self.lattice_neighbors_previous = self.h_lv, weights, nbr_prev
            V = lv.shape[0]
            # relu(cat([aflow, lv]) @ W^T + b) (lm:223-227); alpha = 0 (lm:212, 229)
            if AG.grad_mode():
                lv = torch.relu(torch.nn.functional.linear(torch.cat([aflow_vec, lv], 1), self.linear.weight,
                                                           self.linear.bias))
            else:
              lv = ops.gather_gemm(V, self.linear.weight, ops.gemm_src(aflow_vec), ops.gemm_src(lv), w_is_nk=True,
                                 bias=self.linear.bias, relu=True, stats=True)      # + partial sums for the next Gn
            self.h_lv = _keep(lv)
        ls.set_values(lv)
        return lv, ls


class PointNetSeqModule(torch.nn.Module):
    """reference lm:343-576"""

    def __init__(self, nr_output_channels_per_layer, nr_outputs_last_layer, experiment, rnn_modules,
                 sequence_learning=False, multiplier_hidden_activations=1.0):
        super().__init__()
        self.first_time = True
        self.nr_output_channels_per_layer = nr_output_channels_per_layer
        self.nr_outputs_last_layer = nr_outputs_last_layer
        self.nr_linear_layers = len(self.nr_output_channels_per_layer)
        self.layers = torch.nn.ModuleList([])
        self.norm_layers = torch.nn.ModuleList([])
        self.relu = torch.nn.ReLU(inplace=False)
        self.experiment = experiment
        self.sequence_learning = sequence_learning
        self.h_lv = None
        self.rnn_modules = rnn_modules
        self.multiplier_hidden_activations = multiplier_hidden_activations
        self.fusion_module = None
        c2 = self.nr_output_channels_per_layer[-1] * 2
        kind = rnn_modules[0] if sequence_learning else "none"                       # lm:364-386
        if kind == "linear":
            print("adding Early_Linear fusion with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = TemporalLinearModule(c2)
        elif kind == "cga":
            print("adding Early_CGA with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = CrossframeGlobalAttentionModule(c2)
        elif kind == "aflow":
            print("adding Early_AFLOW with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = CrossframeLocalInterpolationModule(c2)
        elif kind == "lstm":
            print("adding Early_LSTM with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = LSTMModule(c2)
        elif kind == "gru":
            print("adding Early_GRU with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = GRUModule(c2)
        elif kind == "maxpool":
            print("adding Early_MaxPool with nr_output_channels ", self.nr_outputs_last_layer)
            self.fusion_module = TemporalMaxPoolModule()
        self.is_early_maxpool_fusion = (rnn_modules[0] == "maxpool" and sequence_learning)
        self.nr_iters = 0

    def reset_sequence(self):
        if self.fusion_module is not None:
            self.fusion_module.reset_sequence()
        self.h_lv = None

    def _first_time(self, distributed):                                              # lm:410-440
        with torch.no_grad():
            self.first_time = False
            nr_input_channels = distributed.shape[1] - 1
            if self.experiment == "attention_pool":
                nr_input_channels = distributed.shape[1]
            for nr_output_channels in self.nr_output_channels_per_layer:
                self.layers.append(torch.nn.Linear(nr_input_channels, nr_output_channels, bias=True).to("cuda"))
                torch.nn.init.kaiming_normal_(self.layers[-1].weight, mode="fan_in", nonlinearity="relu")
                nr_input_channels = nr_output_channels
            if self.experiment == "attention_pool":
                self.pre_conv = torch.nn.Linear(nr_input_channels, nr_input_channels, bias=False).to("cuda")
                self.gamma = torch.nn.Parameter(torch.ones(nr_input_channels).to("cuda"))
                torch.nn.init.kaiming_normal_(self.pre_conv.weight, mode="fan_in", nonlinearity="relu")
                self.att_activ = GnRelu1x1(nr_input_channels, False)
                self.att_scores = GnRelu1x1(nr_input_channels, True)
            self.last_conv = ConvLatticeModule(nr_filters=self.nr_outputs_last_layer, neighbourhood_size=1,
                                               dilation=1, bias=False)

    def _attention_pool(self, lattice_py, distributed, indices):                     # lm:486-510
        from .compat_scatter import scatter_add, scatter_max
        x = distributed
        for i, layer in enumerate(self.layers):
            x = _linear(layer, x, relu=(i < len(self.layers) - 1))
        indices_long = indices.long()
        indices_long[indices_long < 0] = 0
        V = lattice_py.nr_lattice_vertices()
        max_reduced, _ = scatter_max(x, indices_long, dim=0, dim_size=V)
        x_max = x + self.gamma * torch.index_select(max_reduced, 0, indices_long)
        pre = ops.gather_gemm(x.shape[0], self.pre_conv.weight, ops.gemm_src(x_max), w_is_nk=True)
        rows_ls = _RowStructure()
        att, _ = self.att_activ(pre, rows_ls)
        att, _ = self.att_scores(att, rows_ls)
        att = torch.exp(att)
        att_sum = torch.index_select(scatter_add(att, indices_long, dim=0, dim_size=V), 0, indices_long)
        reduced = scatter_add(x * (att / att_sum), indices_long, dim=0, dim_size=V)
        ones = torch.ones(indices_long.shape[0], 1, device="cuda")
        nr_points = scatter_add(ones, indices_long, dim=0, dim_size=V)
        return reduced.masked_fill(nr_points < 4, 0)

    def forward(self, lattice_py, distributed, indices):
        self.nr_iters += 1
        if self.first_time:
            self._first_time(distributed)
        if self.experiment == "attention_pool":
            distributed_reduced = self._attention_pool(lattice_py, distributed, indices)
        else:
            # lm:448-530 in one fused pass: per-row MLP (skipped for the no-elevation experiments, lm:455-458),
            # scatter_max + argmax, argmax clamp, points-per-vertex, barycentric-of-argmax, <4 mask
            no_elevation = self.experiment in ("pointnet_no_elevate", "pointnet_no_elevate_no_local_mean", "splat")
            layers = [] if no_elevation else list(self.layers)
            pool = AG.pointnet_pool if AG.grad_mode() else ops.pointnet_pool
            distributed_reduced = pool(
                lattice_py, distributed, indices, [l.weight for l in layers], [l.bias for l in layers],
                0 if self.is_early_maxpool_fusion else 4)
        lattice_py.set_values(distributed_reduced)                                   # lm:532

        if self.sequence_learning and self.rnn_modules[0] == "maxpool":              # lm:555-563
            feat_size = distributed_reduced.shape[1]
            if AG.grad_mode():
                rowsum = distributed_reduced[:, 0:int(feat_size / 2)].abs().sum(dim=1).unsqueeze(1)
                distributed_reduced = distributed_reduced.masked_fill(rowsum == 0, -9900)
            else:
                distributed_reduced = ops.fill_empty_rows(distributed_reduced, int(feat_size / 2), -9900.0)
            distributed_reduced, lattice_py = self.fusion_module(distributed_reduced, lattice_py)
        elif self.sequence_learning and self.fusion_module is not None:              # lm:564-565
            distributed_reduced, lattice_py = self.fusion_module(distributed_reduced, lattice_py)

        if distributed_reduced.shape[0] > 0:                                         # lm:569-570
            if AG.grad_mode():
                keep = torch.ones((distributed_reduced.shape[0], 1), device=distributed_reduced.device)
                keep[0] = 0
                distributed_reduced = distributed_reduced * keep
            else:
                if self.sequence_learning and self.fusion_module is not None:
                    distributed_reduced = distributed_reduced.clone()                # the fusion module keeps its own
                distributed_reduced[0, :] = 0
        lattice_py.set_values(distributed_reduced)
        distributed_reduced, lattice_py = self.last_conv(distributed_reduced, lattice_py)   # lm:573
        lattice_py.set_values(distributed_reduced)
        return distributed_reduced, lattice_py


class _RowStructure:
    """stand-in `ls` for modules applied to per-row (not per-vertex) tensors in the attention-pool branch"""

    def set_values(self, t):
        self._v = t
