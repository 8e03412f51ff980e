"""Shared builders for model-level tests (and bench.py): config stand-ins for cfgParser / ModelParams."""
import copy

BASE_MODEL = {
    "positions_mode": "xyz", "values_mode": "reflectance", "pointnet_layers": [16, 32, 64],
    "pointnet_start_nr_channels": 64, "nr_downsamples": 2, "nr_blocks_down_stage": [2, 2, 2],
    "nr_blocks_bottleneck": 3, "nr_blocks_up_stage": [1, 2, 2], "nr_levels_down_with_normal_resnet": 3,
    "nr_levels_up_with_normal_resnet": 3, "compression_factor": 1.0, "dropout_last_layer": 0.0,
    "sequence_learning": True, "rnn_modules": ["gru", "gru", "aflow", "gru"], "train_alpha_beta": True,
    "use_center": False, "experiment": "none",
}


def make_config(rnn_modules=("gru", "gru", "aflow", "gru"), sequence_learning=True, frames=4, sigma=0.6,
                capacity=100000, **model_overrides):
    model = copy.deepcopy(BASE_MODEL)
    model["rnn_modules"] = list(rnn_modules)
    model["sequence_learning"] = sequence_learning
    model.update(model_overrides)
    return {
        "train": {"dataset_name": "semantickitti"},
        "model": model,
        "lattice_gpu": {"hash_table_capacity": capacity, "nr_sigmas": 1, "sigma_0": "%s 3" % sigma},
        "loader_semantic_kitti": {"frames_per_seq": frames, "accumulate_clouds": False, "cloud_scope": 3,
                                  "include_moving_classes": True},
    }


def build_model(contents, nr_classes=26):
    from temporal_latticenet_amd.cfg import cfgParser
    from temporal_latticenet_amd.lattice import ModelParams
    from temporal_latticenet_amd.models import LNN_SEQ
    parser = cfgParser(contents=contents)
    return LNN_SEQ(nr_classes, ModelParams(contents["model"]), parser).to("cuda")


def make_lattice(contents):
    from temporal_latticenet_amd.lattice import Lattice
    lg = contents["lattice_gpu"]
    sigma = float(str(lg["sigma_0"]).split()[0])
    return Lattice.from_params([sigma] * 3, int(lg["hash_table_capacity"]))


def randomize_parameters(model, seed=0):
    """after the lazily created parameters exist: give every tensor a non-trivial value (GroupNorm gamma/beta
    and zero-initialised heads would otherwise hide errors)"""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight"):
                p.copy_((torch.rand(p.shape, generator=g) * 0.5 + 0.75).to(p.device))
            elif name.endswith("norm.bias") or name.endswith(".bias") or name.endswith("bias_ih") or name.endswith("bias_hh"):
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
            elif name.endswith("linear_deltaW.weight"):
                p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(p.device))
            elif name.endswith("alpha") or name.endswith("beta"):
                continue


def oracle_from_model(model, contents, nr_classes=26):
    from oracle.model import OracleLNN
    m = contents["model"]
    lg = contents["lattice_gpu"]
    sigma = float(str(lg["sigma_0"]).split()[0])
    return OracleLNN(model.state_dict(), nr_classes, m["rnn_modules"], m["sequence_learning"], m["pointnet_layers"],
                     m["nr_downsamples"], m["nr_blocks_down_stage"], m["nr_blocks_bottleneck"],
                     m["nr_blocks_up_stage"], [sigma] * 3, int(lg["hash_table_capacity"]), m["experiment"])
