"""Host-side mirror of the reference's seq_lattice/models.py: `LNN_SEQ` with the same constructor, the same
forward signature and return values (models.py:284-476), the same sequence state (`first_sequence`,
`reset_sequence`, models.py:252-263) and the same module / parameter names, on top of the gfx950 operators.

Quirks kept on purpose: the up-path ResNet loop is a SIBLING of the finefy loop and
therefore runs once with i = nr_downsamples-1 (models.py:435-437); the early-return points
(models.py:307, 346, 427); lazily created parameters.
"""
import sys

import torch

from .lattice_modules import (BottleneckBlock, DistributeLatticeModule, GnReluCoarsen, GnReluFinefy, ResnetBlock,
                              SliceFastCUDALatticeModule, SliceLatticeModule, SplatLatticeModule)
from .seq_modules import (CrossframeGlobalAttentionModule, CrossframeLocalInterpolationModule, GRUModule, LSTMModule,
                          PointNetSeqModule, TemporalLinearModule, TemporalMaxPoolModule)

from .checkpoint import load_checkpoint, summary  # noqa: F401  (summary: models.py:551-602, exported by `import *`)

__all__ = ["LNN_SEQ", "make_fusion_module", "summary", "load_checkpoint"]

VALID_EXPERIMENTS = ["none", "slice_no_deform", "pointnet_no_elevate", "pointnet_no_local_mean",
                     "pointnet_no_elevate_no_local_mean", "splat", "attention_pool"]


def make_fusion_module(kind, channels, where=""):
    """the per-depth fusion choices of models.py:76-153"""
    if kind == "linear":
        print("adding %s_Linear fusion with nr_output_channels " % where, channels)
        return TemporalLinearModule(channels)
    if kind == "maxpool":
        print("adding %s_MaxPool fusion" % where)
        return TemporalMaxPoolModule()
    if kind == "cga":
        print("adding %s_CGA fusion with nr_output_channels " % where, channels)
        return CrossframeGlobalAttentionModule(channels)
    if kind == "lstm":
        print("adding %s_LSTM fusion with nr_output_channels " % where, channels)
        return LSTMModule(channels)
    if kind == "gru":
        print("adding %s_GRU fusion with nr_output_channels " % where, channels)
        return GRUModule(channels)
    if kind == "aflow":
        print("adding %s_AFLOW Module with nr_output_channels " % where, channels)
        return CrossframeLocalInterpolationModule(channels)
    return None


class LNN_SEQ(torch.nn.Module):
    def __init__(self, nr_classes, model_params, config_parser):
        print("\n-------- Model definition --------")
        super().__init__()
        self.nr_classes = nr_classes
        model_config = config_parser.get_model_vars()
        loader_config = config_parser.get_loader_vars()
        self.frames_per_seq = loader_config["frames_per_seq"]
        self.multiplier_hidden_activations = 1.0 / self.frames_per_seq

        self.model_params = model_params
        self.nr_downsamples = model_params.nr_downsamples()
        self.nr_blocks_down_stage = model_params.nr_blocks_down_stage()
        self.nr_blocks_bottleneck = model_params.nr_blocks_bottleneck()
        self.nr_blocks_up_stage = model_params.nr_blocks_up_stage()
        self.nr_levels_down_with_normal_resnet = model_params.nr_levels_down_with_normal_resnet()
        self.nr_levels_up_with_normal_resnet = model_params.nr_levels_up_with_normal_resnet()
        compression_factor = model_params.compression_factor()
        dropout_last_layer = model_params.dropout_last_layer()
        experiment = model_params.experiment()
        if experiment not in VALID_EXPERIMENTS:
            sys.exit("Experiment " + experiment + " is not valid")                   # models.py:40-42

        # ---- sequence learning (models.py:48-56)
        self.sequence_learning = model_config["sequence_learning"]
        self.h_lv = None
        self.first_sequence = True
        # True (the reference's contract, models.py:430): an early-return frame hands back the lattice values it stopped
        # at.  A caller that drops them — as train_ln.py / test_ln.py do for every frame but the last — may set this False:
        # the frame program then returns None there instead of copying the values out of its state buffer (the
        # reference returns the module's own tensor: no copy on its side either)
        self.keep_early_values = True
        self.rnn_modules = [x.lower() for x in model_config["rnn_modules"]]
        for i in range(len(self.rnn_modules)):
            if self.rnn_modules[i] not in ["linear", "maxpool", "cga", "aflow", "lstm", "gru"]:
                self.rnn_modules[i] = "none"
        if not loader_config["accumulate_clouds"]:
            print("Fusion Modules: ", self.rnn_modules)
        else:
            print("Accumulating all clouds!")
        assert self.rnn_modules.count("none") < len(self.rnn_modules), \
            "If sequence_learning = True the rnn_modules can not all be none."

        # ---- distribute + PointNet (models.py:62-67)
        self.distribute = DistributeLatticeModule(experiment)
        self.pointnet_layers = model_params.pointnet_layers()
        self.start_nr_filters = model_params.pointnet_start_nr_channels()
        print("pointnet layers is ", self.pointnet_layers)
        self.point_net_seq = PointNetSeqModule(self.pointnet_layers, self.start_nr_filters, experiment,
                                               self.rnn_modules, self.sequence_learning,
                                               self.multiplier_hidden_activations)

        # ---- middle / bottleneck / late fusion (models.py:73-155)
        recurrent_fusion_modules = torch.nn.ModuleList([None] * 3)
        if self.sequence_learning:
            c = self.start_nr_filters
            recurrent_fusion_modules[0] = make_fusion_module(self.rnn_modules[1], c, "Middle")
            recurrent_fusion_modules[1] = make_fusion_module(self.rnn_modules[2], c * 4, "Bottle")
            recurrent_fusion_modules[2] = make_fusion_module(self.rnn_modules[3], c * 3, "Late")
            self.recurrent_fusion_modules = recurrent_fusion_modules

        # ---- down path (models.py:161-184)
        self.resnet_blocks_per_down_lvl_list = torch.nn.ModuleList([])
        self.coarsens_list = torch.nn.ModuleList([])
        self.maxpool_list = torch.nn.ModuleList([])
        skip_connection_channel_counts = []
        cur_channels_count = self.start_nr_filters
        for i in range(self.nr_downsamples):
            self.resnet_blocks_per_down_lvl_list.append(torch.nn.ModuleList([]))
            for j in range(self.nr_blocks_down_stage[i]):
                if i < self.nr_levels_down_with_normal_resnet:
                    print("adding down_resnet_block with nr of filters", cur_channels_count, "and with dropout", False)
                    self.resnet_blocks_per_down_lvl_list[i].append(
                        ResnetBlock(cur_channels_count, [1, 1], [False, False], False))
                else:
                    print("adding down_bottleneck_block with nr of filters", cur_channels_count)
                    self.resnet_blocks_per_down_lvl_list[i].append(
                        BottleneckBlock(cur_channels_count, [False, False, False]))
            skip_connection_channel_counts.append(cur_channels_count)
            nr_channels_after_coarsening = int(cur_channels_count * 2 * compression_factor)
            print("adding bnReluCoarsen which outputs nr of channels ", nr_channels_after_coarsening)
            self.coarsens_list.append(GnReluCoarsen(nr_channels_after_coarsening))
            cur_channels_count = nr_channels_after_coarsening

        # ---- bottleneck (models.py:190-193)
        self.resnet_blocks_bottleneck = torch.nn.ModuleList([])
        for j in range(self.nr_blocks_bottleneck):
            print("adding bottleneck_resnet_block with nr of filters", cur_channels_count)
            self.resnet_blocks_bottleneck.append(BottleneckBlock(cur_channels_count, [False, False, False]))
        self.do_concat_for_vertical_connection = True

        # ---- up path (models.py:201-230)
        self.finefy_list = torch.nn.ModuleList([])
        self.up_activation_list = torch.nn.ModuleList([])
        self.up_match_dim_list = torch.nn.ModuleList([])
        self.up_bn_match_dim_list = torch.nn.ModuleList([])
        self.resnet_blocks_per_up_lvl_list = torch.nn.ModuleList([])
        for i in range(self.nr_downsamples):
            nr_chanels_skip_connection = skip_connection_channel_counts.pop()
            nr_chanels_finefy = int(cur_channels_count / 2)
            print("adding bnReluFinefy which outputs nr of channels ", nr_chanels_finefy)
            self.finefy_list.append(GnReluFinefy(nr_chanels_finefy))
            if self.do_concat_for_vertical_connection:
                cur_channels_count = nr_chanels_skip_connection + nr_chanels_finefy
            else:
                cur_channels_count = nr_chanels_skip_connection
            self.resnet_blocks_per_up_lvl_list.append(torch.nn.ModuleList([]))
            for j in range(self.nr_blocks_up_stage[i]):
                is_last_conv = j == self.nr_blocks_up_stage[i] - 1 and i == self.nr_downsamples - 1
                if i >= self.nr_downsamples - self.nr_levels_up_with_normal_resnet:
                    print("adding up_resnet_block with nr of filters", cur_channels_count)
                    self.resnet_blocks_per_up_lvl_list[i].append(
                        ResnetBlock(cur_channels_count, [1, 1], [False, is_last_conv], False))
                else:
                    print("adding up_bottleneck_block with nr of filters", cur_channels_count)
                    self.resnet_blocks_per_up_lvl_list[i].append(
                        BottleneckBlock(cur_channels_count, [False, False, is_last_conv]))

        self.slice_fast_cuda = SliceFastCUDALatticeModule(nr_classes=self.nr_classes, dropout_prob=dropout_last_layer,
                                                          experiment=experiment)
        self.slice = SliceLatticeModule()
        self.splat = SplatLatticeModule()
        self.start_time = None
        self.logsoftmax = torch.nn.LogSoftmax(dim=1)
        self.lattice_neighbors_previous_index_list, self.avg_position_per_vertex_list, self.weight_vis_list = [], [], []
        if experiment != "none":
            print("-------------------------------\nUSING EXPERIMENT " + experiment + "\n-------------------------------")

    # ---- frame program (engine.py): the same forward as below, as two native calls per inference frame
    use_frame_program = True

    def _fusion_modules_in_state_order(self):
        mods = [self.point_net_seq.fusion_module] + list(self.recurrent_fusion_modules) if self.sequence_learning else []
        return [m for m in mods if m is not None]

    def _program_for_this_frame(self, vis_aflow):
        """decided at the first frame of a sequence; a frame that needs the operator-level route (gradients,
        visualisation hooks) takes the hidden states with it"""
        from . import engine
        wanted = self.use_frame_program and not torch.is_grad_enabled() and not vis_aflow and \
            not getattr(self, "_program_unsupported", False)
        if self.first_sequence or not self.sequence_learning:
            self._program_active = False
            if wanted:
                key = engine.params_key(self)
                if getattr(self, "_program", None) is None or self._program_key != key:
                    key = engine.params_key(self, refresh=True)     # lazily created parameters may have appeared
                    self._program, self._program_key = engine.compile_model(self), key
                prog = self._program
                if prog is not None and not (self.training and prog.uses_dropout):
                    prog.reset()
                    self._program_active = True
        elif getattr(self, "_program_active", False) and not wanted:
            for sid, m in enumerate(self._fusion_modules_in_state_order()):
                m.h_lv = self._program.state(sid)
            self._program_active = False
        return self._program if getattr(self, "_program_active", False) else None

    def reset_sequence(self):                                                        # models.py:252-263
        self.h_lv = None
        self.first_sequence = True
        self._program_active = False
        self.start_time = None
        self.lattice_neighbors_previous_index_list, self.avg_position_per_vertex_list, self.weight_vis_list = [], [], []
        if self.sequence_learning:
            self.point_net_seq.reset_sequence()
            for module in self.recurrent_fusion_modules:
                if module is not None:
                    module.reset_sequence()

    def forward(self, ls, positions, values, early_return=False, with_gradient=True, vis_aflow=False):
        reset_hashmap = True                                                         # models.py:287-289
        if self.sequence_learning and not self.first_sequence:
            reset_hashmap = False
        prog = self._program_for_this_frame(vis_aflow)
        if prog is not None:
            out, ls = prog.run_frame(ls, positions, values, reset_hashmap, early_return, self.keep_early_values)
            self.first_sequence = False
            if early_return and prog.stop_shape is not None:
                return out, out, ls
            fused = prog.take_logsm()          # the slice head wrote log_softmax(scores) beside the scores
            return (fused if fused is not None else self.logsoftmax(out)), out, ls
        with torch.set_grad_enabled(False):
            ls, distributed, indices, weights = self.distribute(ls, positions, values, reset_hashmap)   # :298
            if hasattr(ls, "prepare_levels"):      # all coarse levels + neighbour tables of the frame in one go
                ls.prepare_levels(self.nr_downsamples)
        lv, ls = self.point_net_seq(ls, distributed, indices)                        # :303

        if early_return and self.sequence_learning and self.rnn_modules[1] == "none" and \
                self.rnn_modules[2] == "none" and self.rnn_modules[3] == "none":     # :307-309
            self.first_sequence = False
            return lv, lv, ls

        fine_structures_list = []
        fine_values_list = []
        for i in range(self.nr_downsamples):                                         # :314
            for j in range(self.nr_blocks_down_stage[i]):
                lv, ls = self.resnet_blocks_per_down_lvl_list[i][j](lv, ls)
            fine_structures_list.append(ls)
            fine_values_list.append(lv)
            if i == 0:
                if self.sequence_learning and self.recurrent_fusion_modules[0] is not None:
                    lv, ls = self.recurrent_fusion_modules[0](lv, ls)                # :341-342
                if early_return and self.sequence_learning and self.rnn_modules[2] == "none" and \
                        self.rnn_modules[3] == "none":                               # :346-349
                    self.first_sequence = False
                    return lv, lv, ls
            lv, ls = self.coarsens_list[i](lv, ls)                                   # :353

        for j in range(self.nr_blocks_bottleneck):                                   # :361-363
            lv, ls = self.resnet_blocks_bottleneck[j](lv, ls)
        if self.sequence_learning and self.recurrent_fusion_modules[1] is not None:
            lv, ls = self.recurrent_fusion_modules[1](lv, ls)                        # :381-382

        grad = not (early_return and self.sequence_learning and self.rnn_modules[3] == "none") and with_gradient
        with torch.set_grad_enabled(grad and torch.is_grad_enabled()):               # :386
            for i in range(self.nr_downsamples):                                     # :390
                fine_values = fine_values_list.pop()
                fine_structure = fine_structures_list.pop()
                lv, ls = self.finefy_list[i](lv, ls, fine_structure)                 # :398
                if self.do_concat_for_vertical_connection:
                    lv = torch.cat((lv, fine_values), 1)                             # :401
                else:
                    lv = lv + fine_values
                if i == self.nr_downsamples - 1:
                    if self.sequence_learning and self.recurrent_fusion_modules[2] is not None:
                        lv, ls = self.recurrent_fusion_modules[2](lv, ls)            # :424-425
                    if early_return and self.sequence_learning:                      # :427-430
                        self.first_sequence = False
                        return lv, lv, ls
            # NB: sibling of the loop above, as in the reference (models.py:435-437): runs once, i = last
            for j in range(self.nr_blocks_up_stage[i]):
                lv, ls = self.resnet_blocks_per_up_lvl_list[i][j](lv, ls)

        if vis_aflow:                                                                # :442-461
            self._collect_aflow_vis(lv, ls, positions, indices)

        sv = self.slice_fast_cuda(lv, ls, positions, indices, weights)               # :465
        logsoftmax = self.logsoftmax(sv)
        self.first_sequence = False
        return logsoftmax, sv, ls

    def _collect_aflow_vis(self, lv, ls, positions, indices):
        """models.py:442-461.  The reference reads `self.late_AFLOW`, an attribute that no longer exists
        (SURVEY.md §2.1 row 5); the AFlow module in use is looked up among the fusion slots instead."""
        from .compat_scatter import scatter_mean
        mod = None
        if self.sequence_learning:
            for m in list(self.recurrent_fusion_modules)[::-1] + [self.point_net_seq.fusion_module]:
                if isinstance(m, CrossframeLocalInterpolationModule):
                    mod = m
                    break
        h_lv_vis, weights_vis, nbr_prev = mod.return_for_vis() if mod is not None else (None, None, None)
        if weights_vis is None:
            weights_vis = torch.zeros((lv.shape[0], 1), dtype=torch.long)
            nbr_prev = torch.zeros((lv.shape[0], 1), dtype=torch.long)
        pos_scatter = torch.repeat_interleave(positions, 4, dim=0)
        avg = torch.zeros((lv.shape[0], 3), device="cuda")
        avg = scatter_mean(pos_scatter, indices.clone().type(torch.int64), dim=0, out=avg)
        self.avg_position_per_vertex_list.append(avg.clone())
        self.lattice_neighbors_previous_index_list.append(nbr_prev.clone())
        self.weight_vis_list.append(weights_vis)
        ls.set_values(lv)
        self.first_sequence = False

    def visualize_the_aflow_module(self):                                            # models.py:480-481
        return self.lattice_neighbors_previous_index_list, self.avg_position_per_vertex_list, self.weight_vis_list

    def prepare_cloud(self, cloud):                                                  # models.py:483-531
        with torch.set_grad_enabled(False):
            mode = self.model_params.positions_mode()
            if mode == "xyz":
                positions_tensor = torch.from_numpy(cloud.V).float().to("cuda")
            elif mode == "xyz+rgb":
                positions_tensor = torch.cat((torch.from_numpy(cloud.V).float().to("cuda"),
                                              torch.from_numpy(cloud.C).float().to("cuda")), 1)
            elif mode == "xyz+intensity":
                positions_tensor = torch.cat((torch.from_numpy(cloud.V).float().to("cuda"),
                                              torch.from_numpy(cloud.I).float().to("cuda")), 1)
            else:
                sys.exit("positions mode of " + str(mode) + " not implemented")
            vmode = self.model_params.values_mode()
            if vmode == "none":
                values_tensor = torch.zeros(positions_tensor.shape[0], 1)
            elif vmode == "intensity":
                values_tensor = torch.from_numpy(cloud.I).float().to("cuda")
            elif vmode == "rgb":
                values_tensor = torch.from_numpy(cloud.C).float().to("cuda")
            elif vmode == "rgb+height":
                values_tensor = torch.cat((torch.from_numpy(cloud.C).float().to("cuda"),
                                           torch.from_numpy(cloud.V[:, 1]).unsqueeze(1).float().to("cuda")), 1)
            elif vmode == "rgb+xyz":
                values_tensor = torch.cat((torch.from_numpy(cloud.C).float().to("cuda"),
                                           torch.from_numpy(cloud.V).float().to("cuda")), 1)
            elif vmode == "height":
                values_tensor = torch.from_numpy(cloud.V[:, 1]).unsqueeze(1).float().to("cuda")
            elif vmode == "xyz":
                values_tensor = torch.from_numpy(cloud.V).float().to("cuda")
            else:
                sys.exit("values mode of " + str(vmode) + " not implemented")
            target_tensor = torch.from_numpy(cloud.L_gt).long().squeeze(1).to("cuda").squeeze(0)
        return positions_tensor, values_tensor, target_tensor

    def compute_class_weights(self, class_frequencies, background_idx):              # models.py:535-548
        class_frequencies_tensor = torch.from_numpy(class_frequencies).float().to("cuda")
        class_weights = 1.0 / torch.log(1.05 + class_frequencies_tensor)
        class_weights[background_idx] = 0.00000001
        return class_weights


def forward_group(models, lattices, positions, values, early_return=False):
    """One frame of 2..8 independent sequences in lock-step on the current stream (inference): every gather-GEMM op of
    their frame programs is issued as one launch (engine.FrameProgram.run_frame_group, tln_program_run_group).  Same
    return value per model as LNN_SEQ.forward; falls back to one forward call per model when a model cannot take the
    frame program for this frame."""
    from . import engine
    resets = [not (mod.sequence_learning and not mod.first_sequence) for mod in models]
    progs = [mod._program_for_this_frame(False) for mod in models]
    if len(models) < 2 or len(models) > 8 or any(p is None for p in progs) or any(r != resets[0] for r in resets) or \
            len(set(id(p) for p in progs)) != len(progs):
        return [mod(ls, p, v, early_return, False) for mod, ls, p, v in zip(models, lattices, positions, values)]
    res = engine.FrameProgram.run_frame_group(progs, lattices, positions, values, resets[0], early_return,
                                              any(mod.keep_early_values for mod in models))
    out = []
    for mod, prog, (o, ls) in zip(models, progs, res):
        mod.first_sequence = False
        if early_return and prog.stop_shape is not None:
            out.append((o, o, ls))
        else:
            fused = prog.take_logsm()
            out.append((fused if fused is not None else mod.logsoftmax(o), o, ls))
    return out


forward_pair = forward_group
