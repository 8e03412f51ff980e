"""CPU: host-side logic (cfg reader, ModelParams, model wiring) and the C-ABI library surface."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG_TEXT = '''
core: {
    loguru_verbosity: 3
    hidpi: false  // a comment
}
train: { dataset_name: "semantickitti", lr:0.001
    weight_decay: 1e-3 }
model: {
    positions_mode: "xyz"
    pointnet_layers: [16,32,64]
    nr_blocks_up_stage: [1,2,2]
    compression_factor: 1.0
    //pointnet_layers: [16,32]
    sequence_learning: true
    rnn_modules: ["gru", "gru", "aflow", "gru"] // possibilities are lstm,aflow,...
    experiment: "none"
}
lattice_gpu: {
    hash_table_capacity: 100000 //good for semantic kitti
    nr_sigmas: 1
    sigma_0: "0.6 3" //sigma of X affecting Y dimensions
}
loader_semantic_kitti: { frames_per_seq: 4, accumulate_clouds: false
    label_mngr: { unlabeled_idx: 0 }
    transformer: { hsv_jitter: [0,0,0] } }
'''


def test_cfg_reader_handles_the_reference_dialect():
    from temporal_latticenet_amd.cfg import cfgParser, loads
    c = loads(CFG_TEXT)
    assert c["core"] == {"loguru_verbosity": 3, "hidpi": False}
    assert c["train"]["lr"] == 0.001 and c["train"]["weight_decay"] == 1e-3
    assert c["model"]["pointnet_layers"] == [16, 32, 64]
    assert c["model"]["rnn_modules"] == ["gru", "gru", "aflow", "gru"]
    assert c["lattice_gpu"]["sigma_0"] == "0.6 3" and c["lattice_gpu"]["hash_table_capacity"] == 100000
    p = cfgParser(contents=c)
    assert p.get_loader_vars()["frames_per_seq"] == 4
    assert p.get_label_mngr_vars()["unlabeled_idx"] == 0
    assert p.get_transformer_vars()["hsv_jitter"] == [0, 0, 0]
    assert p.get_model_vars()["sequence_learning"] is True


def test_model_params_getters():
    from temporal_latticenet_amd.cfg import loads
    from temporal_latticenet_amd.lattice import ModelParams
    mp = ModelParams(loads(CFG_TEXT)["model"])
    assert mp.pointnet_layers() == [16, 32, 64] and mp.nr_blocks_up_stage() == [1, 2, 2]
    assert mp.compression_factor() == 1.0 and mp.experiment() == "none" and mp.nr_downsamples() == 2
    assert mp.positions_mode() == "xyz"


def test_model_wiring_matches_the_reference_channel_trace(capsys):
    """models.py:129-153 fusion widths and the module lists, without touching a GPU (parameters are lazy)"""
    from temporal_latticenet_amd.configs import make_config
    from temporal_latticenet_amd.cfg import cfgParser
    from temporal_latticenet_amd.lattice import ModelParams
    from temporal_latticenet_amd.models import LNN_SEQ
    from temporal_latticenet_amd import seq_modules as S
    c = make_config()
    m = LNN_SEQ(26, ModelParams(c["model"]), cfgParser(contents=c))
    assert isinstance(m.point_net_seq.fusion_module, S.GRUModule)
    assert m.point_net_seq.fusion_module.GRU.weight_ih.shape == (3 * 128, 128)
    assert isinstance(m.recurrent_fusion_modules[0], S.GRUModule) and m.recurrent_fusion_modules[0].GRU.hidden_size == 64
    assert isinstance(m.recurrent_fusion_modules[1], S.CrossframeLocalInterpolationModule)
    assert m.recurrent_fusion_modules[1].linear.weight.shape == (256, 512)
    assert m.recurrent_fusion_modules[2].GRU.hidden_size == 192
    assert len(m.resnet_blocks_per_down_lvl_list) == 2 and len(m.resnet_blocks_bottleneck) == 3
    assert [len(x) for x in m.resnet_blocks_per_up_lvl_list] == [1, 2]
    assert m.first_sequence is True
    m.first_sequence = False
    m.reset_sequence()
    assert m.first_sequence is True
    # invalid fusion names fall back to "none"; all-none is rejected (models.py:51-56)
    c2 = make_config(rnn_modules=("foo", "gru", "NONE", "Aflow"))
    m2 = LNN_SEQ(20, ModelParams(c2["model"]), cfgParser(contents=c2))
    assert m2.rnn_modules == ["none", "gru", "none", "aflow"]
    with pytest.raises(AssertionError):
        c3 = make_config(rnn_modules=("x", "y", "z", "w"))
        LNN_SEQ(20, ModelParams(c3["model"]), cfgParser(contents=c3))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tln.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tln_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_library_exports_every_declared_symbol():
    from temporal_latticenet_amd import _lib
    from temporal_latticenet_amd import build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    declared = _declared_symbols()
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, "declared in include/tln.h but not exported: %s" % missing
    # the ctypes prototypes cover exactly the declared surface
    assert sorted(_lib.exported_symbols()) == declared
    assert _lib.lib().tln_version() >= 1            # no device needed


def test_the_library_exports_no_process_wide_switch():
    """"no global state except explicit handles".  Until round 3 the kernel-selection switches were setters
    on file-scope variables (tln_*_config, tln_gemm_force_*), racy under the host threads of streams.py; they are fields of
    tln_options now, stored in a handle or passed with a call.  No such setter is exported, declared or bound, and the
    kernel sources keep no mutable file-scope switch."""
    import subprocess
    from temporal_latticenet_amd import _lib
    gone = ["tln_distribute_config", "tln_distribute_bucket_rows", "tln_pool_config", "tln_gemm_force_tiles",
            "tln_gemm_force_groups", "tln_gemm_force_splits", "tln_gemm_force_direct", "tln_gemm_v2_config",
            "tln_gemm_pair_disable", "tln_gemm_debug_stamps", "tln_program_group_config"]
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    for name in gone:
        assert not hasattr(lib, name), "%s is still exported" % name
        assert name not in declared and name not in _lib.exported_symbols()
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = re.findall(r"\b(tln_[a-z0-9_]+)$", nm.stdout, flags=re.M)
        bad = [e for e in exported if re.search(r"_config$|_force_|_disable$", e)]
        assert not bad, bad
    for want in ("tln_options_init", "tln_lattice_set_options", "tln_program_set_options", "tln_gather_gemm_opt",
                 "tln_gather_gemm_multi_opt", "tln_gru_cell_opt"):
        assert hasattr(lib, want), want
    # no mutable switch at file scope in the kernel sources (immutable `static const` defaults read from the environment
    # once and mutexes are not switches)
    csrc = os.path.join(ROOT, "temporal_latticenet_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            for ln in open(os.path.join(csrc, f)):
                m = re.match(r"^static\s+(?!const\b|constexpr\b|inline\b|thread_local\b)[A-Za-z_0-9:<> \*]+\s+(g_[a-z0-9_]+)\s*(=|;)", ln)
                assert not (m and "mutex" not in ln), "%s: mutable file-scope variable %s" % (f, ln.strip())
    # the defaults as tln_options_init writes them
    o = _lib.Options()
    _lib.lib().tln_options_init(ctypes.byref(o))
    assert o.pool_mode == -1 and all(getattr(o, n) in (0, None) for n, _ in _lib.Options._fields_ if n != "pool_mode")


def test_host_side_options_context():
    """temporal_latticenet_amd/options.py: one struct per host thread, nested blocks, explicit push / set / pop, worker
    threads inherit what run() captured"""
    import threading
    from temporal_latticenet_amd import options as O
    O.reset()
    assert O.current() is None and O.current_ref() is None and O.generation() == 0
    with O.options(v2_min_m=1) as a:
        g1 = O.generation()
        assert a.v2_min_m == 1 and a.pool_mode == -1 and g1 > 0
        with O.options(gemm_direct=-1) as b:
            assert b.v2_min_m == 1 and b.gemm_direct == -1 and O.generation() != g1
        assert O.current() is a and O.generation() == g1
        seen = {}

        def worker(opt, gen):
            seen["before"] = O.current()
            with O.inherit(opt, gen):
                seen["inside"] = (O.current().v2_min_m, O.generation())
            seen["after"] = O.current()

        t = threading.Thread(target=worker, args=(O.current(), O.generation()))
        t.start()
        t.join()
        assert seen == {"before": None, "inside": (1, g1), "after": None}
        O.set(gemm_pair_off=1)
        assert O.current().gemm_pair_off == 1 and O.current().v2_min_m == 1 and O.generation() != g1
    assert O.current() is None
    with pytest.raises(TypeError):
        O.push(no_such_field=1)
    with pytest.raises(RuntimeError):
        O.set(v2_off=1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from temporal_latticenet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TlnError, match="no CPU fallback"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "temporal_latticenet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_synthetic_scan_contract():
    """the loader contract the generator mimics (kitti:100-201): [N,3] f32, [N,1] f32, frames in frame-0 coords"""
    import numpy as np
    from temporal_latticenet_amd.synthetic import make_sequence
    seq = make_sequence(3000, 3, seed=9)
    assert len(seq) == 3
    for pos, val in seq:
        assert pos.shape == (3000, 3) and pos.dtype == np.float32 and val.shape == (3000, 1)
        r = np.linalg.norm(pos, axis=1)
        assert val.min() >= 0 and val.max() < 1
        assert pos[:, 1].min() > -4.5 and pos[:, 1].max() < 6.0      # +y is up, rolling ground around -1.73
    again = make_sequence(3000, 3, seed=9)
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(seq, again)), "seeded"


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """the structs of include/tln.h as gcc lays them out (plain C) against their ctypes mirrors in _lib.py: size and
    the offset of every field, so that a field added on one side only is caught without a GPU"""
    import ctypes as C
    import shutil
    import subprocess
    from temporal_latticenet_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"tln_gemm_src": _lib.GemmSrc, "tln_gemm_call": _lib.GemmCall, "tln_slot": _lib.Slot,
             "tln_op_src": _lib.OpSrc, "tln_op": _lib.Op, "tln_gn_desc": _lib.GnDesc,
             "tln_distribute_call": _lib.DistributeCall, "tln_options": _lib.Options}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tln.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n"):
        if ln.strip():
            a, b, c = ln.split()
            got[(a, b)] = int(c)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, "%s.%s" % (cname, fname)


def test_lattice_scale_constant_cfg_values():
    """lattice_gpu.scale_constant (tln_lattice_create_ex): names and numbers; 0.0 is the C ABI's "default" """
    import pytest
    from temporal_latticenet_amd._lib import TlnError
    from temporal_latticenet_amd.configs import make_config
    from temporal_latticenet_amd.lattice import Lattice
    f = Lattice.parse_scale_constant
    assert f(None) == 0.0 and f("adams") == 0.0 and f("") == 0.0
    assert f("unit") == 1.0 and f(1) == 1.0 and f("1.0") == 1.0 and f(3.25) == 3.25
    with pytest.raises(TlnError):
        f(-1.0)
    with pytest.raises(ValueError):
        f("adam")
    assert "scale_constant" not in make_config()["lattice_gpu"]          # the reference's cfg has no such key
    assert make_config(scale_constant="unit")["lattice_gpu"]["scale_constant"] == "unit"
    from tests.helpers import _scale_constant
    assert _scale_constant(None) is None and _scale_constant("unit") == 1.0 and _scale_constant(2) == 2.0


def test_summary_counts_parameters_and_prints_the_tree():
    """summary() of seq_lattice/models.py:551-602: returns the parameter count, prints one line per module"""
    import io

    import torch
    from temporal_latticenet_amd.checkpoint import summary
    m = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Sequential(torch.nn.Linear(5, 2, bias=False), torch.nn.ReLU()))
    buf = io.StringIO()
    n = summary(m, file=buf)
    assert n == 3 * 5 + 5 + 5 * 2 == sum(p.numel() for p in m.parameters())
    text = buf.getvalue()
    assert "(0): Linear(in_features=3, out_features=5, bias=True), 20 params" in text
    assert text.splitlines()[0].startswith("Sequential(") and text.splitlines()[0].endswith("30 params")
    assert summary(m, file=None) == n
    import temporal_latticenet_amd.models as M
    assert M.summary is summary and "summary" in M.__all__


def test_load_checkpoint_reports_every_key_and_takes_a_rename_map(tmp_path):
    """test_ln.py:174 / train_ln.py:198 call model.load_state_dict(torch.load(path)); the names inside the un-vendored
    modules are unknown here, so the helper reports instead of failing half-way"""
    import pytest
    import torch
    from temporal_latticenet_amd.checkpoint import load_checkpoint

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = torch.nn.GroupNorm(2, 4)
            self.conv = torch.nn.Linear(4, 4)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([Block(), Block()])
            self.head = torch.nn.Linear(4, 3)

    torch.manual_seed(0)
    src, dst = Net(), Net()
    # a checkpoint with upstream-style names: blocks.N.gn.* instead of blocks.N.norm.*, one stray key, one wrong shape
    sd = {k.replace(".norm.", ".gn."): v.clone() for k, v in src.state_dict().items()}
    sd["optimizer_step"] = torch.tensor(3)
    sd["head.weight"] = torch.zeros(5, 4)
    path = str(tmp_path / "model_e_2.pt")
    torch.save(sd, path)
    with pytest.raises(KeyError) as e:
        load_checkpoint(dst, path)
    msg = str(e.value)
    assert "blocks.0.norm.weight" in msg and "blocks.0.gn.weight" in msg and "optimizer_step" in msg and "head.weight" in msg
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    assert all(torch.equal(before[k], v) for k, v in dst.state_dict().items()), "strict failure copies nothing"
    rep = load_checkpoint(dst, path, rename={r"^blocks\.(\d+)\.gn\.": r"blocks.\1.norm."}, strict=False)
    assert rep.unexpected == ["optimizer_step"] and rep.missing == [] and [r[0] for r in rep.shape_mismatch] == ["head.weight"]
    assert len(rep.renamed) == 4 and ("blocks.1.gn.bias", "blocks.1.norm.bias") in rep.renamed
    for k, v in src.state_dict().items():
        if k != "head.weight":
            assert torch.equal(dst.state_dict()[k], v), k
    assert torch.equal(dst.state_dict()["head.weight"], before["head.weight"])
    assert not rep.ok and "1 unexpected" in str(rep)
    # plain prefix form, exact checkpoint: strict passes
    sd2 = {("module." + k): v for k, v in src.state_dict().items()}
    rep = load_checkpoint(Net(), sd2, rename={"module.": ""})
    assert rep.ok and len(rep.loaded) == len(sd2)
