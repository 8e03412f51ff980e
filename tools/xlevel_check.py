import sys; sys.path.insert(0, ".")
import torch, numpy as np
from temporal_latticenet_amd import ops, autograd as AG
from temporal_latticenet_amd.lattice import Lattice
from temporal_latticenet_amd.synthetic import make_sequence
pos, val = make_sequence(20000, 1, seed=21)[0]
lat = Lattice.from_params([0.5] * 3, 1 << 17)
lat.distribute(torch.from_numpy(pos).cuda(), torch.from_numpy(val).cuda())
c = lat.coarsen()
Vf, Vc = lat.nr_lattice_vertices(), c.nr_lattice_vertices()
f2c = c.fine_to_coarse_table(Vf)
print(Vf, Vc, f2c.shape, int((f2c >= 0).sum()))
g = torch.Generator().manual_seed(0)
src = torch.randn(Vc, 64, generator=g).cuda().requires_grad_(True)
W = (torch.randn(9 * 64, 32, generator=g) / 24).cuda().requires_grad_(True)
dout = torch.randn(Vf, 32, generator=g).cuda()
res = []
for mode in ("hip", "hip", "torch"):
    AG.torch_backward(mode == "torch")
    src.grad = None; W.grad = None
    out = AG.gather_gemm(Vf, src, f2c, W, src_lattice=c)
    out.backward(dout)
    res.append((src.grad.clone(), W.grad.clone()))
AG.torch_backward(False)
print("hip twice equal:", torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1]))
print("vs torch: dA", float((res[0][0] - res[2][0]).abs().max()), float(res[2][0].abs().max()), " dW", float((res[0][1] - res[2][1]).abs().max()), float(res[2][1].abs().max()))
