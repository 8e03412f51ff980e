"""GPU: the large-M gather-GEMM (csrc/gemm_v2.hip: 128-row block tiles, operands staged by LDS-DMA) against the
oracle's im2row @ W and against the small/medium-M kernels of gemm.hip on the same inputs.  The kernel normally takes
over at M >= 12288; here it is forced on for a ~9k-vertex lattice (ragged last tile) and also run at its natural size."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import permuto as P
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lattice(gpu):
    from temporal_latticenet_amd.lattice import Lattice
    pos, val = make_sequence(20000, 1, seed=21)[0]
    lat = Lattice.from_params([0.5] * 3, 1 << 17)
    lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    tab = P.VertexTable(3, 1 << 17)
    O.distribute(tab, pos, val, [0.5] * 3)
    assert tab.nr_vertices == lat.nr_lattice_vertices()
    return lat, P.neighbour_table(tab)


@pytest.fixture
def v2_forced():
    """every product of the test on the large-M kernel (kernel-selection options of this host thread, options.py)"""
    from temporal_latticenet_amd import options as OPT
    OPT.push(v2_min_m=1)
    yield OPT
    OPT.pop()


# (cin, cout, taps, weights as [N,K], GroupNorm+ReLU prologue, bias, residual, relu)
CASES = [
    (64, 64, 9, False, True, False, True, False),      # level-0 ResNet conv: GN -> ReLU -> conv + identity
    (128, 64, 9, False, True, False, False, False),    # finefy-shaped
    (192, 192, 9, False, True, True, True, True),      # the K-heavy level-0 product, every epilogue switch
    (256, 128, 9, False, False, True, False, False),   # 128-column tile, no prologue
    (192, 96, 1, True, True, False, False, False),     # slice step-down 1x1 (Linear weights are [out, in])
    (192, 576, 1, True, False, True, False, False),    # GRU projection x @ W_ih^T + b_ih
    (64, 192, 1, True, False, True, False, False),
    (128, 384, 1, True, False, True, False, False),
    (64, 256, 1, False, False, False, False, True),
]


@pytest.mark.parametrize("cin,cout,taps,nk,pro,use_bias,use_res,relu", CASES)
def test_v2_matches_oracle_and_the_other_kernels(gpu, lattice, v2_forced, cin, cout, taps, nk, pro, use_bias, use_res, relu):
    from temporal_latticenet_amd import ops
    lat, table = lattice
    V = lat.nr_lattice_vertices()
    assert V % 128 != 0
    g = torch.Generator().manual_seed(cin * 7 + cout)
    lv = torch.randn(V, cin, generator=g)
    W = torch.randn(taps * cin, cout, generator=g) / np.sqrt(taps * cin)
    bias = torch.randn(cout, generator=g) if use_bias else None
    res = torch.randn(V, cout, generator=g) if use_res else None
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)

    def run():
        kw = {}
        if pro:
            sc, sh = ops.groupnorm_stats(lv.to(gpu), 32, gamma.to(gpu), beta.to(gpu))
            kw = dict(scale=sc, shift=sh, relu=True)
        s0 = ops.gemm_src(lv.to(gpu), lat.neighbour_table_ptr() if taps == 9 else None, taps, **kw)
        Wd = (W.t().contiguous() if nk else W).to(gpu)
        return ops.gather_gemm(V, Wd, s0, w_is_nk=nk, bias=None if bias is None else bias.to(gpu),
                               residual=None if res is None else res.to(gpu), relu=relu, stats=True)

    out = run()
    again = run()
    assert torch.equal(out, again), "bitwise reproducible"
    x = torch.relu(O.group_norm(lv, gamma, beta)) if pro else lv
    want = (O.im2row(x, table) if taps == 9 else x) @ W
    if bias is not None:
        want = want + bias
    if res is not None:
        want = want + res
    if relu:
        want = torch.relu(want)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-4, atol=5e-5)
    # the fused GroupNorm partial sums describe the tensor that was written: one (sum, sum of squares) per group of 32
    # rows and column — the groups follow the product's row order (rows with the same present taps together when the
    # tap table has one, lattice.hip), so what is checked is their number and that together they cover every row once
    st = out._tln_stats.cpu()
    got = out.cpu().double()
    nb = (V + 31) // 32
    assert tuple(st.shape) == (nb, cout, 2)
    np.testing.assert_allclose(st[:, :, 0].sum(0).numpy(), got.sum(0).numpy(), rtol=1e-10, atol=1e-7)
    np.testing.assert_allclose(st[:, :, 1].sum(0).numpy(), (got * got).sum(0).numpy(), rtol=1e-10, atol=1e-7)
    # same product from gemm.hip's kernels (v2 off): equal up to the order of the K summation
    with v2_forced.options(v2_off=1):
        other = run()
    np.testing.assert_allclose(out.cpu().numpy(), other.cpu().numpy(), rtol=1e-4, atol=5e-5)


def test_v2_rows_past_the_source_read_as_zeros(gpu, lattice, v2_forced):
    """the GRU's h @ W_hh: the hidden state has fewer rows than the frame's lattice (lm:59-60 pads with zeros)"""
    from temporal_latticenet_amd import ops
    lat, _ = lattice
    V = lat.nr_lattice_vertices()
    Vh = V - 1234
    g = torch.Generator().manual_seed(3)
    h = torch.randn(Vh, 64, generator=g)
    W = torch.randn(192, 64, generator=g) / 8
    b = torch.randn(192, generator=g)
    s0 = ops.gemm_src(h.to(gpu), None, 1)
    out = ops.gather_gemm(V, W.to(gpu), s0, w_is_nk=True, bias=b.to(gpu))
    want = torch.cat([h, torch.zeros(V - Vh, 64)]) @ W.t() + b
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-5)


def test_v2_at_its_natural_size(gpu):
    """a 120k-point frame at sigma 0.6 (~19k vertices): the library picks the kernel by itself"""
    from temporal_latticenet_amd import ops
    from temporal_latticenet_amd.lattice import Lattice
    pos, val = make_sequence(120000, 1)[0]
    lat = Lattice.from_params([0.6] * 3, 1 << 17)
    lat.distribute(torch.from_numpy(pos).to(gpu), torch.from_numpy(val).to(gpu))
    V = lat.nr_lattice_vertices()
    assert V >= 12288
    nb = lat.neighbour_table().cpu().numpy()
    g = torch.Generator().manual_seed(9)
    lv = torch.randn(V, 192, generator=g)
    W = torch.randn(9 * 192, 192, generator=g) / np.sqrt(9 * 192)
    out = ops.gather_gemm(V, W.to(gpu), ops.gemm_src(lv.to(gpu), lat.neighbour_table_ptr(), 9))
    rows = np.random.default_rng(0).choice(V, 2048, replace=False)
    want = O.im2row(lv, nb[rows]) @ W
    np.testing.assert_allclose(out.cpu().numpy()[rows], want.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cin,cout,taps,nk,pro,nprod", [
    (128, 128, 9, False, True, 3),     # a level-1 convolution of three lock-stepped sequences
    (64, 64, 9, False, False, 2),
    (192, 576, 1, True, False, 2),     # the GRU cell's pair x @ W_ih^T, h @ W_hh^T
    (192, 96, 1, True, True, 4),
    # groups of EIGHT, as bench.py's timed mode issues them (GemmArgsN<8>), the three shape classes its kernel time is
    # dominated by: <4,2,1,2,F,T,2> (128-column tile, two stages), <4,2,1,3,F,T,3> (192 columns), <4,2,1,1,F,T,2> (64)
    (128, 128, 9, False, True, 8),
    (192, 192, 9, False, True, 8),
    (128, 64, 9, False, True, 8),
    (64, 64, 9, False, True, 8),
])
def test_v2_takes_products_whose_rows_only_together_are_large(gpu, lattice, cin, cout, taps, nk, pro, nprod):
    """tln_gather_gemm_multi: products of one shape class with 2.3k-9k rows each (below the large-M kernel's threshold)
    but 12 288 and more together go out as ONE gemm_v2 launch (blockIdx.z = product, the grid sized for the longest).
    Each result against the oracle and bitwise against the same product forced through gemm_v2 alone (same kernel body,
    same tile: the launch they share must not change a bit)."""
    import ctypes as C
    from temporal_latticenet_amd import _lib, ops
    from temporal_latticenet_amd import options as OPT
    from temporal_latticenet_amd.lattice import stream_ptr
    lat, table = lattice
    V = lat.nr_lattice_vertices()
    lib = _lib.lib()
    g = torch.Generator().manual_seed(cin + cout + nprod)
    step = 517 if nprod <= 4 else 211
    Ms = [V // 2 - step * i for i in range(nprod)]         # different row counts, ragged last tiles
    assert min(Ms) >= 1024
    assert sum(Ms) >= 12288 and max(Ms) < 12288
    W = (torch.randn(cout, taps * cin, generator=g) if nk else torch.randn(taps * cin, cout, generator=g)) / np.sqrt(taps * cin)
    Wd = W.to(gpu)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    lvs = [torch.randn(V, cin, generator=g) for _ in range(nprod)]
    bias = torch.randn(cout, generator=g).to(gpu)
    keep, calls, outs = [], (_lib.GemmCall * nprod)(), []
    for i in range(nprod):
        kw = {}
        if pro:
            sc, sh = ops.groupnorm_stats(lvs[i].to(gpu), 32, gamma.to(gpu), beta.to(gpu))
            kw = dict(scale=sc, shift=sh, relu=True)
        src = ops.gemm_src(lvs[i].to(gpu), lat.neighbour_table_ptr() if taps == 9 else None, taps, **kw)
        out = torch.full((Ms[i], cout), float("nan"), device=gpu)
        keep.append((src, kw, out))
        c = calls[i]
        c.M, c.N, c.s0, c.s1 = Ms[i], cout, C.pointer(src[0]), None
        c.d_w, c.w_is_nk, c.d_bias, c.d_residual, c.ld_res, c.relu = Wd.data_ptr(), 1 if nk else 0, bias.data_ptr(), None, 0, 0
        c.d_out, c.ld_out, c.d_stats = out.data_ptr(), cout, None
        outs.append(out)
    _lib.check(lib.tln_gather_gemm_multi(calls, nprod, stream_ptr()), "tln_gather_gemm_multi")
    torch.cuda.synchronize()
    if cout == 128 and pro and not nk:
        # the same shared launch on 64-row tiles (the launch geometry's other choice for this shape class): no bit changes
        first = [o.clone() for o in outs]
        for o in outs:
            o.fill_(float("nan"))
        with OPT.options(v2_off=8):
            _lib.check(lib.tln_gather_gemm_multi_opt(calls, nprod, OPT.current_ref(), stream_ptr()), "tln_gather_gemm_multi_opt")
            torch.cuda.synchronize()
        for i in range(nprod):
            assert torch.equal(outs[i], first[i]), "64-row tiles, product %d" % i
    OPT.push(v2_min_m=1)                                   # the same products one by one through gemm_v2
    try:
        for i in range(nprod):
            src, kw, _ = keep[i]
            alone = ops.gather_gemm(Ms[i], Wd, src, w_is_nk=nk, bias=bias)
            assert torch.equal(outs[i], alone), "product %d" % i
            x = torch.relu(O.group_norm(lvs[i], gamma, beta)) if pro else lvs[i]
            a = (O.im2row(x, table) if taps == 9 else x)[:Ms[i]]
            want = a @ (W.t() if nk else W) + bias.cpu()
            np.testing.assert_allclose(outs[i].cpu().numpy(), want.numpy(), rtol=1e-4, atol=5e-5)
    finally:
        OPT.pop()


@pytest.mark.parametrize("cin,cout,pro", [(64, 64, True), (192, 192, False), (128, 64, True)])
def test_row_order_by_present_taps_changes_no_bit(gpu, lattice, v2_forced, cin, cout, pro):
    """Every 9-tap product over a tap table walks its rows in the table's row order (rows with the same set of present
    neighbour taps together, lattice.hip) and skips the K chunks of the taps no row of a 128-row block has.  A skipped
    chunk would have added exact zeros, and a row's sum does not depend on which rows share its block: the result is
    bitwise the one without the row order (tln_options.v2_off bit 2), and the GroupNorm partial sums still add up to the
    tensor.  On this lattice a third of the neighbour taps is missing: the fixture also checks that there is something
    to skip."""
    from temporal_latticenet_amd import ops
    lat, table = lattice
    V = lat.nr_lattice_vertices()
    missing = float((table[:, :8] < 0).mean())
    assert 0.15 < missing < 0.6, missing
    g = torch.Generator().manual_seed(cin + cout)
    lv = torch.randn(V, cin, generator=g).to(gpu)
    W = (torch.randn(9 * cin, cout, generator=g) / np.sqrt(9 * cin)).to(gpu)
    gamma, beta = (torch.rand(cin, generator=g) + 0.5).to(gpu), torch.randn(cin, generator=g).to(gpu)

    def run():
        kw = {}
        if pro:
            sc, sh = ops.groupnorm_stats(lv, 32, gamma, beta)
            kw = dict(scale=sc, shift=sh, relu=True)
        s0 = ops.gemm_src(lv, lat.neighbour_table_ptr(), 9, **kw)
        return ops.gather_gemm(V, W, s0, stats=True)

    with_order = run()
    with v2_forced.options(v2_off=4):           # large-M kernel on, row order off
        plain = run()
    assert torch.equal(with_order, plain)
    a, b = with_order._tln_stats.double().sum(0), plain._tln_stats.double().sum(0)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-10, atol=1e-7)


def test_products_past_the_buffer_range_take_the_other_kernels(gpu):
    """The large-M kernel addresses source, weights and output through 32-bit buffer offsets (2 GiB each).  A product
    whose source is larger than that must not take it: 8.5M rows x 64 channels = 2.18 GB through a 1x1 product against
    torch on a sample of rows (the kernels with 64-bit addressing compute it)."""
    from temporal_latticenet_amd import ops
    M, cin, cout = 8_500_000, 64, 64
    g = torch.Generator(device=gpu).manual_seed(5)
    x = torch.randn(M, cin, device=gpu, generator=g)
    assert x.numel() * 4 > (1 << 31)
    W = torch.randn(cin, cout, device=gpu, generator=g) / 8.0
    out = ops.gather_gemm(M, W, ops.gemm_src(x))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, 4096, device=gpu), torch.arange(M - 4096, M, device=gpu),
                      torch.randint(0, M, (8192,), device=gpu, generator=g)])
    want = x[rows].double() @ W.double()
    np.testing.assert_allclose(out[rows].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
