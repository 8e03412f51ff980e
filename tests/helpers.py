"""Test helpers: parameter randomisation and the oracle twin of a HIP model."""
from temporal_latticenet_amd.configs import BASE_MODEL, build_model, make_config, make_lattice  # noqa: F401


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


def oracle_from_model(model, contents, nr_classes=26, dtype=None):
    from oracle.model import OracleLNN
    m = contents["model"]
    lg = contents["lattice_gpu"]
    sigma = float(str(lg["sigma_0"]).split()[0])
    return OracleLNN(model.state_dict(), nr_classes, m["rnn_modules"], m["sequence_learning"], m["pointnet_layers"],
                     m["nr_downsamples"], m["nr_blocks_down_stage"], m["nr_blocks_bottleneck"],
                     m["nr_blocks_up_stage"], [sigma] * 3, int(lg["hash_table_capacity"]), m["experiment"],
                     scale_constant=_scale_constant(lg.get("scale_constant")), **({} if dtype is None else {"dtype": dtype}))


def _scale_constant(v):
    """cfg value of lattice_gpu.scale_constant -> what oracle.permuto.scale_factors takes (None = Adams' factor)"""
    if v is None or (isinstance(v, str) and v.strip().lower() in ("", "adams", "default")):
        return None
    if isinstance(v, str) and v.strip().lower() in ("unit", "one", "none"):
        return 1.0
    return float(v)


# parity record of the session (tests/conftest.py::pytest_sessionfinish writes it out)
PARITY = []


def parity_log(what, max_abs, scale, shape=None):
    PARITY.append({"what": what, "max_abs": max_abs, "max_logit": scale, "max_abs_over_max_logit": max_abs / scale,
                   "shape": list(shape) if shape else None})
    print("[parity] %-70s max_abs %.3e  max|logit| %.2f  ratio %.3e" % (what, max_abs, scale, max_abs / scale))
