"""ORACLE (test infrastructure only — never imported by the product path).

PyTorch-CPU eager restatement of LNN_SEQ.forward (reference seq_lattice/models.py:284-476) over the oracle
operators.  It consumes the state_dict of the HIP model, so both sides run the same weights; it is also the
"PyTorch-CPU eager reference" that BASELINE.json asks to time beside the GPU (bench.py `cpu_baseline`, kind "port").
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O
from . import permuto as P

NO_MEAN = ("pointnet_no_local_mean", "pointnet_no_elevate_no_local_mean", "splat")


class Level:
    def __init__(self, capacity):
        self.table = P.VertexTable(3, capacity)
        self.embedded = 0           # fine vertices already embedded (coarse levels)


class OracleLNN:
    def __init__(self, sd, nr_classes, rnn_modules, sequence_learning=True, pointnet_layers=(16, 32, 64),
                 nr_downsamples=2, nr_blocks_down_stage=(2, 2, 2), nr_blocks_bottleneck=3,
                 nr_blocks_up_stage=(1, 2, 2), sigmas=(0.6, 0.6, 0.6), capacity=100000, experiment="none",
                 nr_levels_down_with_normal_resnet=3, nr_levels_up_with_normal_resnet=3, scale_constant=None,
                 dtype=torch.float32):
        # dtype = torch.float64: the reference for GRADIENT checks (tests/test_gpu_train.py) — a float32 CPU autograd pass
        # through ~60 layers carries rounding noise of the size of the errors to be measured
        # dtype = torch.float64 with exact_pool = True: the "truth" reading of the 1e-4 logit bar (tests/test_gpu_fullsize.py,
        # tools/parity64.py) — the pooled PointNet tensor stays the pinned fp32 fma chain (bit-identical on both sides,
        # DESIGN.md section 2), everything behind it runs in float64
        self.dtype = dtype
        self.sd = {k: v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu() for k, v in sd.items()}
        self.nr_classes = nr_classes
        self.rnn = [m if m in ("linear", "maxpool", "cga", "aflow", "lstm", "gru") else "none" for m in rnn_modules]
        self.seq = sequence_learning
        self.pointnet_layers = list(pointnet_layers)
        self.nd = nr_downsamples
        self.down = list(nr_blocks_down_stage)
        self.nbott = nr_blocks_bottleneck
        self.up = list(nr_blocks_up_stage)
        self.sigmas = list(sigmas)
        self.scale_constant = scale_constant     # lattice scale constant (permuto.scale_factors); None = Adams'
        self.capacity = capacity
        self.experiment = experiment
        self.normal_down = nr_levels_down_with_normal_resnet
        self.normal_up = nr_levels_up_with_normal_resnet
        # True: the PointNet MLP in the pinned fma order (bit-exact twin of csrc/pool.hip, O.linear_fma);
        # False: torch's F.linear — the eager CPU path bench.py times as `cpu_baseline`
        self.exact_pool = True
        self.reset_sequence()

    # ---- state ----------------------------------------------------------------------------
    def reset_sequence(self):
        self.levels = [Level(self.capacity) for _ in range(self.nd + 1)]
        self.first = True
        self.h = {}                 # fusion slot -> hidden state

    # ---- building blocks ------------------------------------------------------------------
    def _gn_relu(self, lv, p):
        return torch.relu(O.group_norm(lv, self.sd[p + ".norm.weight"], self.sd[p + ".norm.bias"]))

    def _gnrelu1x1(self, lv, p):
        x = self._gn_relu(lv, p + ".norm")
        return F.linear(x, self.sd[p + ".linear.linear.weight"], self.sd.get(p + ".linear.linear.bias"))

    def _gnreluconv(self, lv, table, p):
        x = self._gn_relu(lv, p + ".norm")
        return O.conv(x, table, self.sd[p + ".conv.weight"], self.sd.get(p + ".conv.bias"))

    def _resnet(self, lv, table, p):
        x = self._gnreluconv(lv, table, p + ".conv1")
        x = self._gnreluconv(x, table, p + ".conv2")
        return x + lv

    def _bottleneck(self, lv, table, p):
        x = self._gnrelu1x1(lv, p + ".contract")
        x = self._gnreluconv(x, table, p + ".conv")
        x = self._gnrelu1x1(x, p + ".expand")
        return x + lv

    def _block(self, lv, table, p):
        return self._resnet(lv, table, p) if (p + ".conv1.conv.weight") in self.sd else self._bottleneck(lv, table, p)

    def _attention_pool(self, dist, indices, v0):
        """reference lm:449, 460-467, 486-510: the MLP runs on the FULL distributed rows (barycentric weight
        included), the per-row features are softmax-weighted per vertex instead of max-pooled"""
        sd, pfx = self.sd, "point_net_seq."
        x = torch.as_tensor(dist)
        nl = len(self.pointnet_layers)
        for i in range(nl):                                                           # lm:460-467
            x = F.linear(x, sd[pfx + "layers.%d.weight" % i], sd[pfx + "layers.%d.bias" % i])
            if i < nl - 1:
                x = torch.relu(x)
        idx = torch.as_tensor(indices).long().clone()
        idx[idx < 0] = 0                                                              # lm:480
        max_reduced, _ = O.scatter_max(x, idx, v0)                                    # lm:488
        x_max = x + sd[pfx + "gamma"] * max_reduced[idx]                              # lm:489-490
        pre = F.linear(x_max, sd[pfx + "pre_conv.weight"])                            # lm:492
        att = self._gnrelu1x1(pre, pfx + "att_activ")                                 # lm:493
        att = torch.exp(self._gnrelu1x1(att, pfx + "att_scores"))                     # lm:494-495
        att_sum = O.scatter_add(att, idx, v0)[idx]                                    # lm:496-497
        reduced = O.scatter_add(x * (att / att_sum), idx, v0)                         # lm:498-501
        nr_points = O.scatter_add(torch.ones(idx.shape[0], 1), idx, v0)               # lm:504-506
        return reduced.masked_fill(nr_points < 4, 0)                                  # lm:507-509

    def _fusion(self, slot, kind, lv, table, p):
        h = self.h.get(slot)
        if kind == "gru":
            new, h = O.gru_step(lv, h, self.sd, p + ".")
        elif kind == "aflow":
            new, h, _ = O.aflow_step(lv, h, table, self.sd, p + ".")
        elif kind == "maxpool":
            if h is None:
                new, h = lv, lv.clone()
            else:
                hp = F.pad(h, (0, 0, 0, lv.shape[0] - h.shape[0]), value=-9999.0)
                new = torch.maximum(hp, lv)
                h = new.clone()
        elif kind == "linear":
            if h is None:
                new, h = lv, lv.clone()
            else:
                h = F.linear(h, self.sd[p + ".hidden_linear.weight"], self.sd[p + ".hidden_linear.bias"])
                hp = F.pad(h, (0, 0, 0, lv.shape[0] - h.shape[0]), value=0)
                new = torch.relu(F.linear(torch.cat([hp, lv], 1), self.sd[p + ".linear.weight"], self.sd[p + ".linear.bias"]))
                h = new.clone()
        elif kind == "lstm":
            if h is None:
                new, h = lv, lv.clone()
            else:
                h = F.linear(h, self.sd[p + ".hidden_linear.weight"], self.sd[p + ".hidden_linear.bias"])
                hp = F.pad(h, (0, 0, 0, lv.shape[0] - h.shape[0]), value=0)
                g = F.linear(lv, self.sd[p + ".lstm.weight_ih"], self.sd[p + ".lstm.bias_ih"]) + \
                    F.linear(hp, self.sd[p + ".lstm.weight_hh"], self.sd[p + ".lstm.bias_hh"])
                i, f, gg, o = g.chunk(4, 1)
                c = torch.sigmoid(i) * torch.tanh(gg)
                new = torch.sigmoid(o) * torch.tanh(c)
                h = new.clone()
        elif kind == "cga":                                                           # lm:84-116
            if h is None:
                new, h = lv, lv.clone()
            else:
                vh = h.shape[0]
                h = F.linear(h, self.sd[p + ".hidden_linear.weight"], self.sd[p + ".hidden_linear.bias"])
                x = F.pad(h, (0, 0, 0, lv.shape[0] - vh), value=0)
                w = self.sd[p + ".conv.linear.weight"]
                x = torch.relu(F.linear(x, w))
                x = O.group_norm(x, self.sd[p + ".groupnorm.norm.weight"], self.sd[p + ".groupnorm.norm.bias"])
                x = F.linear(x, w)
                x = torch.sigmoid(x * (1.0 / (x.shape[0] + x.shape[1])))
                x[vh:] = 1.0
                new = x * lv
                h = new.clone()
        else:
            return lv
        self.h[slot] = h
        return new

    # ---- forward --------------------------------------------------------------------------
    def forward(self, positions, values, early_return=False):
        sd = self.sd
        positions = np.asarray(positions, np.float32)
        values = np.asarray(values, np.float32)
        if not (self.seq and not self.first):                                         # models.py:287-289
            for lvl in self.levels:
                lvl.table.clear()
                lvl.embedded = 0
        l0 = self.levels[0]
        dist, indices, weights = O.distribute(l0.table, positions, values, self.sigmas,
                                              self.experiment not in NO_MEAN, self.scale_constant)   # models.py:298
        v0 = l0.table.nr_vertices
        tables = [P.neighbour_table(l0.table)]
        # PointNetSeq (lm:407-576)
        nl = len(self.pointnet_layers)
        if self.experiment in ("pointnet_no_elevate", "pointnet_no_elevate_no_local_mean", "splat"):
            ws, bs = [], []
        else:
            ws = [sd["point_net_seq.layers.%d.weight" % i] for i in range(nl)]
            bs = [sd["point_net_seq.layers.%d.bias" % i] for i in range(nl)]
        early_maxpool = self.seq and self.rnn[0] == "maxpool"
        if self.experiment == "attention_pool":
            lv = self._attention_pool(dist, indices, v0)
        else:
            lv = O.pointnet_pool(dist, indices, v0, ws, bs, 0 if early_maxpool else 4, exact=self.exact_pool)
            lv = lv.to(self.dtype)
        if self.seq and self.rnn[0] == "maxpool":
            rowsum = lv[:, : lv.shape[1] // 2].abs().sum(1, keepdim=True)
            lv = lv.masked_fill(rowsum == 0, -9900)
        if self.seq:
            lv = self._fusion("early", self.rnn[0], lv, tables[0], "point_net_seq.fusion_module")
        lv = lv.clone()
        lv[0, :] = 0                                                                  # lm:569-570
        lv = O.conv(lv, tables[0], sd["point_net_seq.last_conv.weight"])              # lm:573
        if early_return and self.seq and self.rnn[1] == "none" and self.rnn[2] == "none" and self.rnn[3] == "none":
            self.first = False
            return lv

        skips = []
        level = 0
        for i in range(self.nd):
            for j in range(self.down[i]):
                lv = self._block(lv, tables[level], "resnet_blocks_per_down_lvl_list.%d.%d" % (i, j))
            skips.append((level, lv))
            if i == 0:
                if self.seq:
                    lv = self._fusion("middle", self.rnn[1], lv, tables[level], "recurrent_fusion_modules.0")
                if early_return and self.seq and self.rnn[2] == "none" and self.rnn[3] == "none":
                    self.first = False
                    return lv
            # coarsen (models.py:353)
            fine, coarse = self.levels[level], self.levels[level + 1]
            P.coarsen_insert(coarse.table, fine.table.keys[coarse.embedded:])
            coarse.embedded = fine.table.nr_vertices
            c2f = fine.table.lookup(P.neighbour_keys(coarse.table.keys * 2))
            p = "coarsens_list.%d" % i
            x = self._gn_relu(lv, p + ".norm")
            lv = O.im2row(x, c2f) @ sd[p + ".coarse.weight"]
            level += 1
            tables.append(P.neighbour_table(coarse.table))

        for j in range(self.nbott):
            lv = self._bottleneck(lv, tables[level], "resnet_blocks_bottleneck.%d" % j)
        if self.seq:
            lv = self._fusion("bottle", self.rnn[2], lv, tables[level], "recurrent_fusion_modules.1")

        for i in range(self.nd):
            flevel, fine_values = skips.pop()
            fine, coarse = self.levels[flevel], self.levels[flevel + 1]
            f2c = coarse.table.lookup(P.neighbour_keys(P.finefy_centres(fine.table.keys)))
            p = "finefy_list.%d" % i
            x = self._gn_relu(lv, p + ".norm")
            lv = O.im2row(x, f2c) @ sd[p + ".fine.weight"]
            lv = torch.cat((lv, fine_values), 1)
            level = flevel
            if i == self.nd - 1:
                if self.seq:
                    lv = self._fusion("late", self.rnn[3], lv, tables[level], "recurrent_fusion_modules.2")
                if early_return and self.seq:
                    self.first = False
                    return lv
        for j in range(self.up[i]):                                                   # models.py:435-437 (sibling loop)
            lv = self._block(lv, tables[level], "resnet_blocks_per_up_lvl_list.%d.%d" % (i, j))

        # slice (models.py:465)
        p = "slice_fast_cuda"
        delta = None
        if self.experiment != "slice_no_deform":
            b = lv
            for k in range(2):
                b = self._gnrelu1x1(b, p + ".stepdown.%d" % k)
            b = self._gnrelu1x1(b, p + ".bottleneck")
            g = O.slice_gather(b, indices, weights)
            hdn = torch.relu(F.linear(g, sd[p + ".linear_pre_deltaW.weight"]))
            delta = F.linear(hdn, sd[p + ".linear_deltaW.weight"], sd[p + ".linear_deltaW.bias"]).reshape(-1)
        feat = O.slice_blend(lv, indices, weights, delta)
        sv = F.linear(feat, sd[p + ".linear_clasify.weight"], sd[p + ".linear_clasify.bias"])
        self.first = False
        return sv
