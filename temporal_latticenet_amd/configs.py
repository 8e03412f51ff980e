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
                capacity=100000, scale_constant=None, **model_overrides):
    model = copy.deepcopy(BASE_MODEL)
    model["rnn_modules"] = list(rnn_modules)
    model["sequence_learning"] = sequence_learning
    model.update(model_overrides)
    lg_extra = {} if scale_constant is None else {"scale_constant": scale_constant}
    return {
        "train": {"dataset_name": "semantickitti"},
        "model": model,
        "lattice_gpu": {"hash_table_capacity": capacity, "nr_sigmas": 1, "sigma_0": "%s 3" % sigma, **lg_extra},
        "loader_semantic_kitti": {"frames_per_seq": frames, "accumulate_clouds": False, "cloud_scope": 3,
                                  "include_moving_classes": True},
    }


def suggest_capacity(nr_points, sigma, frames=1):
    """hash_table_capacity (cfg:71 makes it a hand-set knob: "100000 // good for semantic kitti which splat around 10k
    with sigma of 1") from the cloud size and the lattice scale instead.  Calibration: a 120k-point 64-beam scan hashes
    to ~19k vertices at sigma = 0.6 and ~8.6k at sigma = 1.0 (vertices ~ sigma^-1.6: surfaces, not volumes), a sensor
    moving 1.5 m per frame adds ~20 % of that per further frame; a sparser sampling of the same scene still touches most
    of its vertices (~N^0.2 below 120k points), more points add vertices sub-linearly (~N^0.8 above).  Three times that estimate, never more than the hard bound 4 * points * frames (every point touches at
    most d+1 = 4 vertices; the bound is what very fine lattices reach), never less than 4096."""
    n = float(nr_points)
    rel = n / 120000.0
    est = 19000.0 * rel ** (0.8 if rel > 1.0 else 0.2) * (0.6 / float(sigma)) ** 1.6 * (1.0 + 0.25 * (frames - 1))
    hard = 4.0 * n * frames
    cap = int(min(3.0 * est, hard))
    return max(4096, (cap + 1023) // 1024 * 1024)


def build_model(contents, nr_classes=26):
    from .cfg import cfgParser
    from .lattice import ModelParams
    from .models import LNN_SEQ
    parser = cfgParser(contents=contents)
    return LNN_SEQ(nr_classes, ModelParams(contents["model"]), parser).to("cuda")


def make_lattice(contents, nr_points=None, frames=None):
    """hash_table_capacity: a number, or "auto" -> suggest_capacity(nr_points per frame, sigma, frames of the sequence)"""
    from .lattice import Lattice
    lg = contents["lattice_gpu"]
    sigma = float(str(lg["sigma_0"]).split()[0])
    cap = lg["hash_table_capacity"]
    if isinstance(cap, str) and cap.strip().lower() == "auto":
        if nr_points is None:
            raise ValueError('hash_table_capacity "auto" needs the number of points per frame')
        if frames is None:
            frames = int(contents.get("loader_semantic_kitti", {}).get("frames_per_seq", 1))
        cap = suggest_capacity(nr_points, sigma, frames)
    return Lattice.from_params([sigma] * 3, int(cap), scale_constant=lg.get("scale_constant", None))
