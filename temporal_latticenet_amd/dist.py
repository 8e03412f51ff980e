"""Multi-GPU layer: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on the MI355X
node; "gloo" for the CPU rehearsals in tests/).  The reference is single-process / single-GPU, so
nothing here mirrors reference code; it must only preserve the single-GPU sequential semantics.

Two ways to use N GPUs:

 * sequence sharding (default of bench.py): a sequence owns its lattice and hidden states (train_ln.py:236-239),
   so ranks take disjoint sequences and never exchange data — `shard_items`, `max_over_ranks`.

 * frame sharding: rank g owns frame-slot g of every sequence.  Vertex numbering must equal the sequential one,
   so the ranks all-gather the first-touch-ordered NEW KEYS of their frames (`all_gather_rows`) and insert the
   frames before theirs in frame order (`Lattice.insert_keys`); the recurrence is honoured by handing each fusion
   module's hidden state [V_s, C_s] from rank g to rank g+1 (`send_tensor` / `recv_tensor`, point-to-point: one
   xGMI link) right after it is produced, which turns a stream of sequences into a systolic pipeline.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "shard_items", "max_over_ranks", "all_gather_rows", "send_tensor", "recv_tensor",
           "FrameShardPlan"]


def init_from_env(backend=None, device_index=None):
    """RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device_index is not None:
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_items(n_items, rank, world):
    """items (sequences) owned by `rank`: strided so that every rank gets the same count +-1"""
    return list(range(rank, n_items, world))


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's step time is the slowest rank's)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_rows(rows, group=None):
    """all-gather of 2-D tensors with a different number of rows per rank -> list (rank order).
    Used for the per-frame new-vertex keys [dV_g, 3] int32."""
    world = dist.get_world_size(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def send_tensor(t, dst, tag=0, group=None):
    """shape header + payload (hidden states grow from frame to frame, so the receiver cannot know V_s)"""
    hdr = torch.tensor([t.dim()] + list(t.shape) + [0] * (4 - t.dim()), dtype=torch.int64, device=t.device)
    dist.send(hdr, dst, group=group, tag=tag)
    dist.send(t.contiguous(), dst, group=group, tag=tag)


def recv_tensor(src, device, dtype=torch.float32, tag=0, group=None):
    hdr = torch.zeros(5, dtype=torch.int64, device=device)
    dist.recv(hdr, src, group=group, tag=tag)
    h = hdr.tolist()
    shape = h[1:1 + h[0]]
    t = torch.empty(shape, dtype=dtype, device=device)
    dist.recv(t, src, group=group, tag=tag)
    return t


class FrameShardPlan:
    """Which frames of a T-frame sequence a rank owns, who precedes / follows it, and which ranks form its group.

    world <= T : one group, rank g owns the contiguous block of T/world frames [g*T/world, (g+1)*T/world)
    world >  T : world/T independent groups of T ranks, one frame each (groups take different sequences)
    """

    def __init__(self, nr_frames, rank, world):
        if world <= nr_frames:
            if nr_frames % world:
                raise ValueError("frames (%d) must be a multiple of the ranks (%d)" % (nr_frames, world))
            self.group_size, self.nr_groups = world, 1
        else:
            if world % nr_frames:
                raise ValueError("ranks (%d) must be a multiple of the frames (%d)" % (world, nr_frames))
            self.group_size, self.nr_groups = nr_frames, world // nr_frames
        self.nr_frames = nr_frames
        self.group = rank // self.group_size
        self.slot = rank % self.group_size
        per = nr_frames // self.group_size
        self.frames = list(range(self.slot * per, (self.slot + 1) * per))
        base = self.group * self.group_size
        self.group_ranks = list(range(base, base + self.group_size))
        self.prev_rank = base + self.slot - 1 if self.slot > 0 else None
        self.next_rank = base + self.slot + 1 if self.slot < self.group_size - 1 else None

    def owns_last_frame(self):
        return self.frames[-1] == self.nr_frames - 1
