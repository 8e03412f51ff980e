"""GPU parity of the whole hot path: LNN_SEQ (HIP) vs the CPU oracle on the same weights.
Tolerance: per-point logits within 1e-4 (fp32), the bar BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch

from tests.helpers import build_model, make_config, make_lattice, oracle_from_model, randomize_parameters
from temporal_latticenet_amd import options as OPT
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4


def _run(model, contents, seq, gpu):
    lat = make_lattice(contents)
    outs = []
    with torch.no_grad():
        for t, (pos, val) in enumerate(seq):
            early = t != len(seq) - 1
            a, b, lat = model(lat, torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu), early, False)
            outs.append(b.cpu())
    model.reset_sequence()
    return outs


@pytest.mark.parametrize("rnn", [("gru", "gru", "aflow", "gru"), ("gru", "gru", "gru", "gru"),
                                 ("maxpool", "linear", "lstm", "aflow"), ("cga", "none", "gru", "linear")])
def test_sequence_logits_match_oracle(gpu, rnn):
    contents = make_config(rnn_modules=rnn, frames=3, sigma=0.6)
    seq = make_sequence(15000, 3, seed=31)
    model = build_model(contents).eval()
    _run(model, contents, seq, gpu)                 # creates the lazily built parameters (train_ln.py:175-209)
    randomize_parameters(model, seed=1)
    outs = _run(model, contents, seq, gpu)
    oracle = oracle_from_model(model, contents)
    for t, (pos, val) in enumerate(seq):
        want = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        got = outs[t]
        assert got.shape == want.shape
        scale = max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= LOGIT_TOL * scale, "frame %d: max abs err %.3e (scale %.2f)" % (t, err, scale)
    assert outs[-1].shape == (15000, 26)


@pytest.mark.parametrize("scale_constant", ["unit", 2.0])
def test_sequence_logits_match_oracle_under_another_lattice_scale_constant(gpu, scale_constant):
    """cfg key lattice_gpu.scale_constant (Lattice.create / make_lattice -> tln_lattice_create_ex): the whole path with the
    lattice scale constant an upstream build WITHOUT Adams' (d+1) sqrt(2/3) would use ("unit": cells 3.27x wider in
    every direction), and with a third value -- against the oracle run with the same constant"""
    contents = make_config(rnn_modules=("gru", "gru", "aflow", "gru"), frames=3, sigma=0.3, scale_constant=scale_constant)
    seq = make_sequence(15000, 3, seed=37)
    model = build_model(contents).eval()
    _run(model, contents, seq, gpu)
    randomize_parameters(model, seed=2)
    from temporal_latticenet_amd.configs import make_lattice as mk
    lat = mk(contents)
    assert abs(lat.scale_constant() - (1.0 if scale_constant == "unit" else scale_constant)) < 1e-12
    outs = _run(model, contents, seq, gpu)
    oracle = oracle_from_model(model, contents)
    assert oracle.scale_constant == (1.0 if scale_constant == "unit" else scale_constant)
    for t, (pos, val) in enumerate(seq):
        want = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        scale = max(1.0, float(want.abs().max()))
        err = float((outs[t] - want).abs().max())
        assert err <= LOGIT_TOL * scale, "frame %d: max abs err %.3e (scale %.2f)" % (t, err, scale)
    assert oracle.levels[0].table.nr_vertices > 500


def test_single_frame_no_sequence_learning(gpu):
    contents = make_config(rnn_modules=("gru", "none", "none", "none"), sequence_learning=False, frames=1, sigma=1.0)
    seq = make_sequence(20000, 1, seed=2)
    model = build_model(contents, nr_classes=20).eval()
    _run(model, contents, seq, gpu)
    randomize_parameters(model, seed=2)
    outs = _run(model, contents, seq, gpu)
    oracle = oracle_from_model(model, contents, 20)
    want = oracle.forward(*seq[0])
    err = float((outs[0] - want).abs().max())
    assert err <= LOGIT_TOL * max(1.0, float(want.abs().max())), err


def test_state_dict_roundtrip_and_lazy_parameters(gpu):
    contents = make_config(frames=2)
    seq = make_sequence(8000, 2, seed=4)
    model = build_model(contents).eval()
    n_before = len(model.state_dict())
    outs = _run(model, contents, seq, gpu)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert len(sd) > n_before                      # parameters appear at the first forward (lm:288-295, 410-440)
    # the reference's checkpoint-visible names
    for k in ["point_net_seq.layers.0.weight", "point_net_seq.fusion_module.GRU.weight_ih",
              "point_net_seq.fusion_module.hidden_linear.weight", "recurrent_fusion_modules.1.AFLOW.alpha",
              "recurrent_fusion_modules.1.AFLOW.weight", "recurrent_fusion_modules.1.linear.weight",
              "recurrent_fusion_modules.2.GRU.bias_hh"]:
        assert k in sd, k
    assert tuple(sd["recurrent_fusion_modules.1.AFLOW.weight"].shape) == (9 * 256, 256)
    # up-level-0 blocks never run (models.py:435-437) => they never get conv weights
    assert not any(k.startswith("resnet_blocks_per_up_lvl_list.0.") and k.endswith("conv.weight") for k in sd)
    model2 = build_model(contents).eval()
    _run(model2, contents, seq, gpu)
    model2.load_state_dict(sd)
    outs2 = _run(model2, contents, seq, gpu)
    assert torch.equal(outs[-1], outs2[-1])        # same weights => bitwise identical (deterministic kernels)


CFG = """
train: { dataset_name: "semantickitti" }
model: {
    positions_mode: "xyz"
    values_mode: "reflectance"
    pointnet_layers: [16,32,64]
    pointnet_start_nr_channels: 64
    nr_downsamples: 2
    nr_blocks_down_stage: [2,2,2]
    nr_blocks_bottleneck: 3
    nr_blocks_up_stage: [1,2,2]
    nr_levels_down_with_normal_resnet: 3
    nr_levels_up_with_normal_resnet: 3
    compression_factor: 1.0
    dropout_last_layer: 0.0
    sequence_learning: true
    rnn_modules: ["gru", "gru", "aflow", "gru"] // the pretrained configuration
    experiment: "none"
}
lattice_gpu: {
    hash_table_capacity: 100000 //good for semantic kitti
    nr_sigmas: 1
    sigma_0: "0.6 3"
}
loader_semantic_kitti: { frames_per_seq: 2, accumulate_clouds: false, include_moving_classes: true }
"""


def test_reference_import_names_drive_the_model(gpu, tmp_path):
    """the reference's own import lines (train_ln.py:15-29, models.py:6-12) resolve to this implementation and the
    per-frame calling convention of train_ln.py:160-239 works: Lattice.create(cfg), ModelParams.create(cfg),
    model(lattice, positions, values, early_return, with_gradient), reset_sequence(), a fresh Lattice per sequence"""
    import temporal_latticenet_amd
    temporal_latticenet_amd.install_compat()
    cfg = tmp_path / "lnn.cfg"
    cfg.write_text(CFG)
    ns = {}
    exec("from easypbr import *\nfrom latticenet import ModelParams, Lattice, HashTable\n"
         "from latticenet_py.lattice.lovasz_loss import LovaszSoftmax\nfrom latticenet_py.lattice.lattice_funcs import *\n"
         "from latticenet_py.lattice.lattice_modules import *\nimport torch_scatter\nimport hjson\n"
         "from termcolor import colored\nfrom seq_lattice.models import *\n", ns)
    from temporal_latticenet_amd.cfg import cfgParser
    config_parser = cfgParser(str(cfg))
    assert ns["hjson"].loads(CFG)["lattice_gpu"]["sigma_0"] == "0.6 3"
    model_params = ns["ModelParams"].create(str(cfg))
    lattice = ns["Lattice"].create(str(cfg), "lattice")
    model = ns["LNN_SEQ"](26, model_params, config_parser).to("cuda")
    loss_fn = ns["LovaszSoftmax"](ignore_index=0)
    seq = make_sequence(6000, 2, seed=8)
    target = torch.randint(0, 26, (6000,)).to(gpu)
    for epoch in range(2):
        for i, (pos, val) in enumerate(seq):
            with torch.set_grad_enabled(False):
                early = i != len(seq) - 1
                pred, raw, lattice = model(lattice, torch.from_numpy(pos).to("cuda"), torch.from_numpy(val).to("cuda"),
                                           early, with_gradient=False)
        loss = 0.5 * loss_fn(pred, target) + 0.5 * torch.nn.NLLLoss(ignore_index=0)(pred, target)
        assert torch.isfinite(loss)
        assert lattice.nr_lattice_vertices() > 100
        model.reset_sequence()
        lattice = ns["Lattice"].create(str(cfg), "lattice")
    assert pred.shape == (6000, 26)
    mx, arg = ns["torch_scatter"].scatter_max(torch.rand(50, 3, device="cuda"), torch.randint(0, 7, (50,), device="cuda"), dim=0)
    assert mx.shape == (7, 3)


@pytest.mark.parametrize("experiment,rnn", [
    ("slice_no_deform", ("gru", "gru", "aflow", "gru")),
    ("pointnet_no_local_mean", ("gru", "gru", "aflow", "gru")),
    ("pointnet_no_elevate", ("none", "gru", "none", "gru")),
    ("pointnet_no_elevate_no_local_mean", ("none", "gru", "none", "gru")),
    ("splat", ("none", "gru", "none", "gru")),
    ("attention_pool", ("none", "gru", "none", "gru")),
])
def test_experiments_match_oracle(gpu, experiment, rnn):
    """the `experiment` switches of models.py:39 (PointNet without elevation / local mean, attention pooling
    lm:486-510, slice without the deformation head); the second pass runs through the frame program where the
    configuration supports it"""
    contents = make_config(rnn_modules=rnn, frames=2, sigma=0.7, experiment=experiment)
    seq = make_sequence(8000, 2, seed=41)
    model = build_model(contents).eval()
    _run(model, contents, seq, gpu)
    randomize_parameters(model, seed=3)
    outs = _run(model, contents, seq, gpu)
    oracle = oracle_from_model(model, contents)
    for t, (pos, val) in enumerate(seq):
        want = oracle.forward(pos, val, early_return=(t != len(seq) - 1))
        got = outs[t]
        assert got.shape == want.shape
        scale = max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= LOGIT_TOL * scale, "%s frame %d: max abs err %.3e (scale %.2f)" % (experiment, t, err, scale)


def test_executed_flops_of_a_frame_are_counted_by_the_kernels(gpu):
    """bench.py's roofline.executed: the kernels' own step counters over one extra pass of a frame's products.  With the
    large-M kernel's tap skipping the executed work is below the algorithmic 2*M*K*N (the im2row product's zero rows),
    without it (every product through the small-M kernels) it is at least the algorithmic figure (tile padding), and the
    counting pass leaves the results alone."""
    from temporal_latticenet_amd import _lib
    from temporal_latticenet_amd.engine import FrameProgram
    contents = make_config(frames=2, sigma=0.6)
    seq = make_sequence(60000, 2, seed=5)
    model = build_model(contents).eval()
    _run(model, contents, seq, gpu)                 # creates the lazily built parameters
    _run(model, contents, seq, gpu)                 # compiles the frame program
    prog = model._program
    assert prog is not None
    shares = {}
    for name, force in (("default", 0), ("small-M kernels only", 1)):
        OPT.push(gemm_direct=force)
        try:
            prog.capture_gemms(True)
            lat = make_lattice(contents)
            with torch.no_grad():
                a, first, lat = model(lat, torch.from_numpy(seq[0][0]).to(gpu), torch.from_numpy(seq[0][1]).to(gpu), False, False)
                ms, n, fl, by = prog.replay_gemms(1)
                ex = FrameProgram.replay_executed([prog])
                again = model._program.replay_gemms(1)
            prog.capture_gemms(False)
            model.reset_sequence()
        finally:
            OPT.pop()
        assert n > 20 and fl > 0 and again[2] == fl
        shares[name] = ex / fl
    print("executed / algorithmic flops:", shares)
    assert 0.5 < shares["default"] < 0.98
    assert 1.0 <= shares["small-M kernels only"] < 1.1
