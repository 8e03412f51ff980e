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


def parity_log(what, max_abs, scale, shape=None, **more):
    rec = {"what": what, "max_abs": max_abs, "max_logit": scale, "max_abs_over_max_logit": max_abs / scale,
           "shape": list(shape) if shape else None}
    rec.update(more)
    PARITY.append(rec)
    extra = "".join("  %s %.3e" % (k, v) for k, v in more.items() if isinstance(v, float))
    print("[parity] %-70s max_abs %.3e  max|logit| %.2f  ratio %.3e%s" % (what, max_abs, scale, max_abs / scale, extra))


# The north star's tolerance: per-point logits "within 1e-4 fp32" of the reference path.
NORTH_STAR_TOL = 1e-4


def oracle_pair(model, contents, **kw):
    """(fp32 oracle, float64 oracle) on the same weights.  The float64 one keeps the pooled PointNet tensor in the pinned
    fp32 fma order (bit-identical to the HIP pool, oracle/model.py) and runs everything behind it in float64: the
    reference point that tells the rounding noise of the HIP path from that of the fp32 CPU restatement."""
    import torch
    return oracle_from_model(model, contents, **kw), oracle_from_model(model, contents, dtype=torch.float64, **kw)


MAX_SLACK = 1.25   # the worst of ~3 M elements is an extreme-value statistic: two evaluations with the SAME noise level differ
                   # in it by this much from draw to draw (tools/parity64.py: 0.8-1.4 between HIP and o32 over the BASELINE cases)
RMS_SLACK = 1.15   # the noise LEVEL itself: rms of (x - o64) over all elements


def float64_everywhere():
    """TLN_TEST_FLOAT64=all: the float64 oracle also runs in the longest full-size cases (the 8 x 120k and 8 x 30k
    recurrences, the 1.0M-vertex cloud, three of the four sequences of the timed configuration, vis_aflow) — +7 minutes of
    CPU time on the GPU box;
    profiles/r04_parity_errors.json was recorded that way.  Default: those cases keep the fp32 comparison and the float64
    reading is asserted on the 4 x 120k sequences, config 2, one sequence of the timed configuration, the accumulated
    clouds at sigma 0.6 and the smaller recurrences."""
    import os
    return os.environ.get("TLN_TEST_FLOAT64", "") == "all"


def check_logits(got, want32, want64, what, rel_tol=NORTH_STAR_TOL, abs_ceiling=None):
    """The 1e-4 bar settled with a float64 reference.  Recorded per comparison: the max-abs and the rms
    of HIP - o64, o32 - o64 (and max-abs HIP - o32).  Asserted:
      * max|HIP - o64| <= max(1e-4, 1.25 x max|o32 - o64|): the HIP path is within the north star's absolute 1e-4 of the
        float64 truth, or — where an fp32 evaluation of ~60 layers cannot be, at |logit| ~ 30-100 — no further from it
        than the fp32 CPU restatement of the same algorithm is (up to the draw-to-draw spread of a maximum);
      * rms(HIP - o64) <= 1.15 x rms(o32 - o64): the NOISE LEVEL of the HIP path is that of the fp32 CPU restatement (round
        3's kernels: 1.55 x, one accumulation chain over all of K in gemm_v2; round 4 folds per tap group: 1.0-1.08 x);
      * |HIP - o32| <= 1e-4 * max(1, max|logit|): the relative reading of earlier rounds, kept so the records stay
        comparable; `abs_ceiling` (when given) bounds the absolute error as a regression tripwire."""
    import torch
    got = got.detach().cpu()
    if want64 is None:          # (a case that runs the fp32 oracle only, see float64_everywhere)
        assert got.shape == want32.shape
        scale = max(1.0, float(want32.abs().max()))
        e_h32 = float((got - want32).abs().max())
        parity_log(what, e_h32, scale, tuple(got.shape), hip_vs_o32=e_h32)
        assert e_h32 <= rel_tol * scale, "%s: |HIP - o32| = %.3e (scale %.2f)" % (what, e_h32, scale)
        if abs_ceiling is not None:
            assert e_h32 <= abs_ceiling, "%s: max abs err %.3e exceeds the recorded absolute level" % (what, e_h32)
        return e_h32
    assert got.shape == want32.shape == want64.shape, (got.shape, want32.shape, want64.shape)
    scale = max(1.0, float(want64.abs().max()))
    d_h, d_o = got.double() - want64, want32.double() - want64
    e_h64, e_3264 = float(d_h.abs().max()), float(d_o.abs().max())
    r_h64, r_3264 = float(d_h.pow(2).mean().sqrt()), float(d_o.pow(2).mean().sqrt())
    e_h32 = float((got - want32).abs().max())
    parity_log(what, e_h32, scale, tuple(got.shape), hip_vs_o64=e_h64, o32_vs_o64=e_3264, hip_vs_o32=e_h32,
               hip_vs_o64_rms=r_h64, o32_vs_o64_rms=r_3264)
    assert e_h64 <= max(NORTH_STAR_TOL, MAX_SLACK * e_3264), \
        "%s: |HIP - o64| = %.3e exceeds max(1e-4, %.2f x |o32 - o64| = %.3e) (max|logit| %.1f)" % (what, e_h64, MAX_SLACK, e_3264, scale)
    assert r_h64 <= RMS_SLACK * r_3264 + 1e-9, \
        "%s: rms(HIP - o64) = %.3e exceeds %.2f x rms(o32 - o64) = %.3e" % (what, r_h64, RMS_SLACK, r_3264)
    assert e_h32 <= rel_tol * scale, "%s: |HIP - o32| = %.3e (scale %.2f)" % (what, e_h32, scale)
    if abs_ceiling is not None:
        assert e_h32 <= abs_ceiling, "%s: max abs err %.3e exceeds the recorded absolute level" % (what, e_h32)
    return e_h64
