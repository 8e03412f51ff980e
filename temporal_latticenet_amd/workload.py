"""The sequences bench.py times, built in one place so that the parity test of the timed configuration
(tests/test_gpu_fullsize.py::test_the_timed_configuration_*) and bench.py's own `checked` field run EXACTLY what the
timed region runs: one ray-cast drive per sequence stream (a different seed each: different vertex counts) and, for
the further sequences of a stream's lock-step group, that drive turned about the vertical axis (again other vertex
counts, without paying the CPU ray casting 32 times).  Stands in for the batches train_ln.py:160-239 / test_ln.py:133-264
take from the SemanticKITTI loader."""
import math

import torch

from .synthetic import make_sequence

__all__ = ["turned", "stream_drives", "group_sequences"]


def turned(drive, j):
    """the drive (list of (positions [N,3], values [N,1]) device tensors) turned by 0.7*j rad about +y (the loader's up axis)"""
    if j == 0:
        return drive
    c, s_ = math.cos(0.7 * j), math.sin(0.7 * j)
    rot = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]], device=drive[0][0].device)
    return [((p @ rot.T).contiguous(), v) for p, v in drive]


def stream_drives(points, frames, seed, streams, first=None, device="cuda"):
    """one drive per stream; `first` (already on the device) is reused as stream 0's when given"""
    drives = [first] if first is not None else []
    for i in range(len(drives), streams):
        drives.append([(torch.from_numpy(p).to(device), torch.from_numpy(v).to(device))
                       for p, v in make_sequence(points, frames, seed=seed + 1000 * i)])
    return drives


def group_sequences(drives, per):
    """flat list, stream-major: sequence j of stream i at [per * i + j]"""
    return [turned(drives[i], j) for i in range(len(drives)) for j in range(per)]
