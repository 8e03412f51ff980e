"""CPU rehearsal of the N>1 paths with world_size 2 over gloo (the GPU runs use the same code over RCCL).

 * sequence sharding: disjoint strided ownership, MAX-over-ranks timing
 * frame sharding: all-gather of first-touch-ordered new keys -> identical vertex numbering on every rank, and
   the hidden-state hand-off rank g -> g+1 reproduces the single-process sequential result exactly (checked
   with the CPU oracle model standing in for the HIP model: the protocol is device independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ops as O
from oracle import permuto as P
from oracle.model import OracleLNN
from temporal_latticenet_amd import dist as D
from temporal_latticenet_amd.synthetic import make_sequence

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, *args, world=WORLD):
    port = _free_port()
    mp.spawn(_entry, args=(fn, port, world) + args, nprocs=world, join=True)


def _entry(rank, fn, port, world, *args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2 if world <= 2 else 1)
    D.init_from_env("gloo")
    try:
        fn(rank, *args)
    finally:
        try:
            dist.destroy_process_group()
        except Exception:          # (a test that let a receive time out leaves a torn-down gloo pair behind)
            pass


# ---------------------------------------------------------------------------------------------
def _sequence_sharding(rank):
    mine = D.shard_items(7, rank, WORLD)
    got = [None] * WORLD
    dist.all_gather_object(got, mine)
    flat = sorted(x for g in got for x in g)
    assert flat == list(range(7)) and abs(len(got[0]) - len(got[1])) <= 1
    t = D.max_over_ranks(1.0 + rank)
    assert t == 2.0
    # uneven all-gather
    rows = torch.arange((rank + 1) * 3 * 3, dtype=torch.int32).reshape(-1, 3) + 100 * rank
    allr = D.all_gather_rows(rows)
    assert [a.shape[0] for a in allr] == [3, 6]
    assert torch.equal(allr[rank], rows)
    # point-to-point with a shape header
    if rank == 0:
        D.send_tensor(torch.full((5, 7), 3.5), 1, tag=4)
    else:
        r = D.recv_tensor(0, "cpu", tag=4)
        assert r.shape == (5, 7) and float(r.sum()) == 3.5 * 35


def test_sequence_sharding_and_primitives():
    _spawn(_sequence_sharding)


def _handoff_out_of_step(rank):
    """the point-to-point hand-off is matched by ORDER on RCCL (tags ignored): every message carries its number on the
    (src -> dst) channel and the fusion slot the sender serves, and the receiver checks both"""
    # in step: three messages, the third one empty (a state that does not exist yet)
    if rank == 0:
        for slot, t in enumerate([torch.ones(3, 4), torch.zeros(2, 2), torch.zeros(0)]):
            D.send_tensor(t, 1, tag=slot)
    else:
        for slot, shape in enumerate([(3, 4), (2, 2), (0,)]):
            assert tuple(D.recv_tensor(0, "cpu", tag=slot).shape) == shape
    dist.barrier()
    # the sender serves slot 5, the receiver waits for slot 6 (same tag on the wire, as RCCL would deliver it)
    if rank == 0:
        ch = D._channel(None, 1)
        hdr = torch.tensor([D._MAGIC, D._sent[ch], 5, 1, 2, 0, 0, 0], dtype=torch.int64)
        dist.send(hdr, 1, tag=6)          # (not counted: the receiver rejects it, and counts completed messages only)
    else:
        with pytest.raises(D.HandoffError, match="out of step"):
            D.recv_tensor(0, "cpu", tag=6)
    dist.barrier()
    # a message number that does not follow (a message was lost or sent twice)
    if rank == 0:
        dist.send(torch.tensor([D._MAGIC, 99, 7, 1, 2, 0, 0, 0], dtype=torch.int64), 1, tag=7)
    else:
        with pytest.raises(D.HandoffError, match="expected message"):
            D.recv_tensor(0, "cpu", tag=7)
    dist.barrier()
    # steady state of the frame-sharded program route: the receiver knows the shape, the payload travels alone (one message
    # per state, no header to read back on the host) -- and the message counters still advance in step
    if rank == 0:
        before = D._sent[D._channel(None, 1)]
        D.send_tensor(torch.full((5, 3), 2.0), 1, tag=9, header=False)
        assert D._sent[D._channel(None, 1)] == before + 1
        D.send_tensor(torch.full((2, 3), 4.0), 1, tag=10)                 # with header, shape checked against the expectation
    else:
        got = D.recv_tensor(0, "cpu", tag=9, shape=(5, 3))
        assert tuple(got.shape) == (5, 3) and float(got.sum()) == 30.0
        assert tuple(D.recv_tensor(0, "cpu", tag=10).shape) == (2, 3)
    dist.barrier()
    # nobody sends: the receiver gives up instead of hanging the pipeline (last: gloo tears the pair down on a timeout)
    if rank == 1:
        with pytest.raises(D.HandoffError):
            D.recv_tensor(0, "cpu", tag=8, timeout_s=1.0)
    else:
        import time
        time.sleep(3.0)


def test_handoff_messages_are_numbered_and_time_out():
    _spawn(_handoff_out_of_step)


# ---------------------------------------------------------------------------------------------
def _tiny_state_dict(seed=0):
    """random weights for a small LNN (1 downsample, GRU early + AFlow bottleneck + GRU late) in the key layout
    of the HIP model's state_dict"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.2
    sd = {}
    dims = [4, 8, 16]
    for i in range(2):
        sd["point_net_seq.layers.%d.weight" % i] = r(dims[i + 1], dims[i])
        sd["point_net_seq.layers.%d.bias" % i] = r(dims[i + 1])
    c0 = 16  # start channels

    def gru(p, c):
        sd[p + ".GRU.weight_ih"], sd[p + ".GRU.weight_hh"] = r(3 * c, c), r(3 * c, c)
        sd[p + ".GRU.bias_ih"], sd[p + ".GRU.bias_hh"] = r(3 * c), r(3 * c)
        sd[p + ".hidden_linear.weight"], sd[p + ".hidden_linear.bias"] = r(c, c), r(c)

    def gn(p, c):
        sd[p + ".norm.weight"], sd[p + ".norm.bias"] = torch.rand(c, generator=g) + 0.5, r(c)

    def gnconv(p, cin, cout):
        gn(p + ".norm", cin)
        sd[p + ".conv.weight"] = r(9 * cin, cout)

    def g1x1(p, cin, cout):
        gn(p + ".norm", cin)
        sd[p + ".linear.linear.weight"] = r(cout, cin)

    gru("point_net_seq.fusion_module", 32)
    sd["point_net_seq.last_conv.weight"] = r(9 * 32, c0)
    gnconv("resnet_blocks_per_down_lvl_list.0.0.conv1", c0, c0)
    gnconv("resnet_blocks_per_down_lvl_list.0.0.conv2", c0, c0)
    gn("coarsens_list.0.norm", c0)
    sd["coarsens_list.0.coarse.weight"] = r(9 * c0, 2 * c0)
    p = "resnet_blocks_bottleneck.0"
    g1x1(p + ".contract", 32, 8)
    gnconv(p + ".conv", 8, 8)
    g1x1(p + ".expand", 8, 32)
    p = "recurrent_fusion_modules.1"
    sd[p + ".AFLOW.alpha"], sd[p + ".AFLOW.beta"] = torch.tensor(0.1), torch.tensor(0.1)
    sd[p + ".AFLOW.bias"] = r(32)
    sd[p + ".linear.weight"], sd[p + ".linear.bias"] = r(32, 64), r(32)
    gn("finefy_list.0.norm", 32)
    sd["finefy_list.0.fine.weight"] = r(9 * 32, 16)
    gru("recurrent_fusion_modules.2", 32)
    gnconv("resnet_blocks_per_up_lvl_list.0.0.conv1", 32, 32)
    gnconv("resnet_blocks_per_up_lvl_list.0.0.conv2", 32, 32)
    p = "slice_fast_cuda"
    g1x1(p + ".stepdown.0", 32, 32)
    g1x1(p + ".stepdown.1", 32, 16)
    g1x1(p + ".bottleneck", 16, 8)
    sd[p + ".linear_pre_deltaW.weight"] = r(36, 36)
    sd[p + ".linear_deltaW.weight"], sd[p + ".linear_deltaW.bias"] = r(4, 36) * 0.1, r(4) * 0.1
    sd[p + ".linear_clasify.weight"], sd[p + ".linear_clasify.bias"] = r(5, 32), r(5)
    return sd


def _make_oracle():
    return OracleLNN(_tiny_state_dict(), 5, ["gru", "none", "aflow", "gru"], True, pointnet_layers=[8, 16],
                     nr_downsamples=1, nr_blocks_down_stage=[1], nr_blocks_bottleneck=1, nr_blocks_up_stage=[1],
                     sigmas=[0.9] * 3, capacity=1 << 14)


SLOTS = ["early", "middle", "bottle", "late"]


def _frame_sharding(rank, WORLD=WORLD):
    seq = make_sequence(1500, WORLD, seed=17)
    plan = D.FrameShardPlan(WORLD, rank, WORLD)
    assert plan.frames == [rank] and plan.group_ranks == list(range(WORLD))
    pos, val = seq[rank]
    # 1. first-touch-ordered keys of my frame alone
    scratch = P.VertexTable(3, 1 << 14)
    O.distribute(scratch, pos, val, [0.9] * 3)
    keys = D.all_gather_rows(torch.from_numpy(scratch.keys))
    # 2. my model sees the frames before mine as already-inserted vertices
    model = _make_oracle()
    for k in keys[:rank]:
        model.levels[0].table.insert(k.numpy())
    if rank > 0:
        model.first = False
        for s in SLOTS:                                  # 3. hidden states from the previous frame's owner
            flag = D.recv_tensor(plan.prev_rank, "cpu", tag=1)
            if flag.numel():
                model.h[s] = flag
    out = model.forward(pos, val, early_return=not plan.owns_last_frame())
    if plan.next_rank is not None:
        for s in SLOTS:
            h = model.h.get(s)
            D.send_tensor(h if h is not None else torch.zeros(0), plan.next_rank, tag=1)
    # reference: the plain sequential run
    ref = _make_oracle()
    for t, (p_, v_) in enumerate(seq):
        want = ref.forward(p_, v_, early_return=(t != WORLD - 1))
        if t == rank:
            break
    assert model.levels[0].table.nr_vertices == ref.levels[0].table.nr_vertices
    assert np.array_equal(model.levels[0].table.keys, ref.levels[0].table.keys), "numbering equals the sequential one"
    assert np.array_equal(model.levels[1].table.keys, ref.levels[1].table.keys)
    assert torch.equal(out, want), "frame-sharded result is bit-identical to the sequential one"
    if plan.owns_last_frame():
        assert out.shape == (1500, 5)


def test_frame_sharding_equals_sequential_semantics():
    _spawn(_frame_sharding)


def test_frame_sharding_over_four_ranks():
    """BASELINE config 4's cut: a 4-frame sequence, one frame per rank (key all-gather over the four, hidden states handed
    from rank g to g + 1 three times): numbering and last-frame scores bit-identical to the sequential run"""
    _spawn(_frame_sharding, 4, world=4)
