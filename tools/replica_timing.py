import time, sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import torch
import bench
from tests.helpers import build_model, make_config, make_lattice
from temporal_latticenet_amd.synthetic import make_sequence
from temporal_latticenet_amd.streams import share_parameters
contents = make_config(rnn_modules=("gru","gru","aflow","gru"), frames=4, sigma=0.6)
seq = [(torch.from_numpy(p).cuda(), torch.from_numpy(v).cuda()) for p, v in make_sequence(120000, 4, seed=1)]
q = contextlib.redirect_stdout(io.StringIO())
with q, torch.no_grad():
    base = build_model(contents).eval(); lat = make_lattice(contents)
    for t,(p,v) in enumerate(seq): base(lat, p, v, t != 3, False)
    base.reset_sequence()
torch.cuda.synchronize()
for rep in range(2):
    t0=time.perf_counter()
    with q, torch.no_grad():
        m = build_model(contents).eval()
    t1=time.perf_counter()
    with q, torch.no_grad():
        lat = make_lattice(contents)
    t2=time.perf_counter()
    with q, torch.no_grad():
        for t,(p,v) in enumerate(seq):
            m(lat, p[:4096], v[:4096], t != 3, False)
        m.reset_sequence(); torch.cuda.synchronize()
    t3=time.perf_counter()
    m = share_parameters(m, base)
    t4=time.perf_counter()
    with q, torch.no_grad():
        for t,(p,v) in enumerate(seq):
            m(lat, p, v, t != 3, False)
        m.reset_sequence(); torch.cuda.synchronize()
    t5=time.perf_counter()
    print("build %.2f lattice %.2f warm %.2f share %.2f first-full-seq %.2f" % (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4), file=sys.stderr)
