"""GPU: the frames of ONE sequence in flight together on one device (pipeline.FramePipeline: a slot per frame — stream,
host thread, model replica, lattice — hidden states and vertex keys handed over inside the process).  The outputs must
be BITWISE those of the sequential route (the same frame program ops, identical vertex numbering), also for sequences
that follow one another through the slots."""
import contextlib
import io

import pytest
import torch

from tests.helpers import build_model, make_config, make_lattice, randomize_parameters
from temporal_latticenet_amd.pipeline import FramePipeline
from temporal_latticenet_amd.synthetic import make_sequence

pytestmark = pytest.mark.gpu


def _sequential(model, contents, seq):
    lat = make_lattice(contents)
    with torch.no_grad():
        for t, (p, v) in enumerate(seq):
            a, b, lat = model(lat, p, v, t != len(seq) - 1, False)
    model.reset_sequence()
    return a.clone(), b.clone()


@pytest.mark.parametrize("rnn,points", [(("gru", "gru", "aflow", "gru"), 20000), (("maxpool", "linear", "lstm", "aflow"), 9000)])
def test_pipelined_frames_compute_what_the_sequential_route_computes(gpu, rnn, points):
    T = 4
    contents = make_config(rnn_modules=rnn, frames=T, sigma=0.6, capacity=1 << 18)
    seqs = [[(torch.from_numpy(p).to(gpu), torch.from_numpy(v).to(gpu)) for p, v in make_sequence(points, T, seed=70 + s)]
            for s in range(3)]
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(3)
        model = build_model(contents).eval()
        _sequential(model, contents, seqs[0])          # creates the lazily built parameters
        randomize_parameters(model, 11)
    want = [_sequential(model, contents, s) for s in seqs]
    with FramePipeline(model, lambda: build_model(contents).eval(), lambda: make_lattice(contents), seqs[0]) as pipe:
        got = pipe.run(seqs[:1])                       # one sequence alone: the latency path
        torch.cuda.synchronize()
        assert torch.equal(got[0][1], want[0][1]) and torch.equal(got[0][0], want[0][0])
        got = pipe.run(seqs)                           # three sequences following one another through the slots
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(got[i][1], want[i][1]), "sequence %d: raw scores differ" % i
            assert torch.equal(got[i][0], want[i][0]), "sequence %d: log-softmax differs" % i
    # the base model still runs the sequential route afterwards
    again = _sequential(model, contents, seqs[1])
    assert torch.equal(again[1], want[1][1])
