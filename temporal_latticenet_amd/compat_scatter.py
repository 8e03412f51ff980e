"""torch_scatter 2.0.4 call surface used by the reference (lattice_modules.py:485-520, models.py:454),
served by the gfx950 scatter kernels.  Only dim=0 is supported (the only form the reference uses)."""
import torch

from . import ops

__all__ = ["scatter_max", "scatter_add", "scatter_mean"]


def _prep(src, index, dim, dim_size, out):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("only dim=0 scatter is implemented")
    squeeze = src.dim() == 1
    src2 = src.reshape(src.shape[0], -1)
    if dim_size is None:
        dim_size = out.shape[0] if out is not None else (int(index.max()) + 1 if index.numel() else 0)
    return src2, squeeze, int(dim_size)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    src2, squeeze, n = _prep(src, index, dim, dim_size, out)
    res, arg = ops.scatter_max(src2, index, n)
    shape = (n,) + tuple(src.shape[1:])
    return res.reshape(shape), arg.reshape(shape)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    src2, squeeze, n = _prep(src, index, dim, dim_size, out)
    o2 = out.reshape(n, -1) if out is not None else None
    res = ops.scatter_add(src2, index, n, o2)
    return res.reshape((n,) + tuple(src.shape[1:]))


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    src2, squeeze, n = _prep(src, index, dim, dim_size, out)
    total = scatter_add(src, index, dim, out, n)
    ones = torch.ones((src.shape[0], 1), dtype=torch.float32, device=src.device)
    cnt = ops.scatter_add(ones, index, n).clamp_(min=1)
    cnt = cnt.reshape((n,) + (1,) * (src.dim() - 1))
    total /= cnt
    return total
