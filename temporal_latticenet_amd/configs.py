"""Config builders: the reference's pretrained configuration (seq_config/lnn_train_semantic_kitti.cfg:31-75) as a
dict, plus constructors for LNN_SEQ / Lattice from it (used by bench.py, __graft_entry__.py and the tests)."""
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
    from .cfg import cfgParser
    from .lattice import ModelParams
    from .models import LNN_SEQ
    parser = cfgParser(contents=contents)
    return LNN_SEQ(nr_classes, ModelParams(contents["model"]), parser).to("cuda")


def make_lattice(contents):
    from .lattice import Lattice
    lg = contents["lattice_gpu"]
    sigma = float(str(lg["sigma_0"]).split()[0])
    return Lattice.from_params([sigma] * 3, int(lg["hash_table_capacity"]))
