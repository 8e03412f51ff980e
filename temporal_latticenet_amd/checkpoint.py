"""Checkpoint helpers around the reference's `model.load_state_dict(torch.load(model_path))` (test_ln.py:174,
train_ln.py:198 — called AFTER the first forward pass, because the un-vendored modules create their parameters lazily).

The parameter names of the in-tree modules are the reference's (`point_net_seq.layers.*`, `*.fusion_module.GRU.*`,
`*.AFLOW.{alpha,beta,weight,bias}`, `*.linear.*`); the names INSIDE the un-vendored modules (ResnetBlock, GnReluCoarsen,
the slice head) cannot be recovered from the reference tree.  `load_checkpoint` therefore never
fails silently and never half-loads: it applies a rename map, reports every key that is missing, unexpected or of another
shape, and copies nothing unless the caller accepts the report (strict=False) or the report is empty.

    report = load_checkpoint(model, "model_e_2.pt", rename={r"^resnet_blocks_(.*)\\.gn1\\.": r"resnet_blocks_\\1.conv1.norm.norm."})
    print(report)          # loaded / missing / unexpected / shape_mismatch / renamed
"""
import re
import sys
from functools import reduce

import torch

__all__ = ["CheckpointReport", "load_checkpoint", "summary"]


class CheckpointReport:
    def __init__(self):
        self.loaded, self.missing, self.unexpected, self.shape_mismatch, self.renamed = [], [], [], [], []

    @property
    def ok(self):
        return not (self.missing or self.unexpected or self.shape_mismatch)

    def __str__(self):
        lines = ["checkpoint: %d tensors loaded, %d renamed, %d missing, %d unexpected, %d of another shape"
                 % (len(self.loaded), len(self.renamed), len(self.missing), len(self.unexpected), len(self.shape_mismatch))]
        for title, rows in (("missing in the checkpoint (model keeps its value)", self.missing),
                            ("unexpected in the checkpoint (not used)", self.unexpected)):
            if rows:
                lines.append("  %s:" % title)
                lines += ["    " + r for r in rows]
        if self.shape_mismatch:
            lines.append("  shape mismatch (checkpoint vs model):")
            lines += ["    %s: %s vs %s" % r for r in self.shape_mismatch]
        if self.renamed:
            lines.append("  renamed:")
            lines += ["    %s -> %s" % r for r in self.renamed[:20]]
            if len(self.renamed) > 20:
                lines.append("    ... %d more" % (len(self.renamed) - 20))
        return "\n".join(lines)


def _apply_renames(key, rename):
    for pat, rep in rename:
        if isinstance(pat, str) and not any(ch in pat for ch in "^$\\(["):
            if key.startswith(pat):                      # plain prefix
                return rep + key[len(pat):]
        else:
            new, n = re.subn(pat, rep, key)
            if n:
                return new
    return key


def load_checkpoint(model, checkpoint, rename=None, strict=True, map_location="cpu"):
    """model: any torch.nn.Module whose lazily created parameters exist already (run one forward first, as test_ln.py does).
    checkpoint: a path (torch.load) or a state dict.  rename: dict or list of (prefix | regex, replacement), first match
    wins, applied to the CHECKPOINT's keys.  strict: raise KeyError with the full report unless every key matches;
    strict=False copies what matches and returns the report."""
    sd = torch.load(checkpoint, map_location=map_location) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "read") \
        else checkpoint
    if isinstance(sd, dict) and "state_dict" in sd and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd["state_dict"]
    rename = list(rename.items()) if isinstance(rename, dict) else list(rename or [])
    own = model.state_dict()
    rep = CheckpointReport()
    incoming = {}
    for k, v in sd.items():
        nk = _apply_renames(k, rename)
        if nk != k:
            rep.renamed.append((k, nk))
        if nk in incoming:
            raise KeyError("the rename map sends two checkpoint keys to %r" % nk)
        incoming[nk] = v
    good = {}
    for k, v in incoming.items():
        if k not in own:
            rep.unexpected.append(k)
        elif tuple(v.shape) != tuple(own[k].shape):
            rep.shape_mismatch.append((k, tuple(v.shape), tuple(own[k].shape)))
        else:
            good[k] = v
    rep.missing = [k for k in own if k not in incoming]
    if strict and not rep.ok:
        raise KeyError(str(rep))
    with torch.no_grad():
        for k, v in good.items():
            own[k].copy_(v)                 # state_dict() tensors alias the parameters / buffers
            rep.loaded.append(k)
    if hasattr(model, "_program_key"):      # a frame program bakes parameter values in (engine.params_key notices writes)
        model._program_key = None
    return rep


def summary(self, file=sys.stderr):
    """`summary(model)` of seq_lattice/models.py:551-602 (exported by `from seq_lattice.models import *`, train_ln.py:29):
    the module tree, one line per module, with the number of parameters below every node; returns the total.  Lazily
    created parameters count once they exist.  file=None only counts."""
    def walk(mod, indent):
        own = sum(reduce(lambda a, b: a * b, p.shape, 1) for p in mod._parameters.values() if p is not None)
        lines, total = [], own
        for name, child in mod._modules.items():
            if child is None:
                continue
            sub, n = walk(child, indent + 2)
            head = "%s(%s): %s" % (" " * (indent + 2), name, sub[0].lstrip())
            lines += [head] + sub[1:]
            total += n
        extra = mod.extra_repr()
        first = "%s%s(%s), %s params" % (" " * indent, mod._get_name(), extra, format(total, ","))
        return [first] + lines, total

    lines, count = walk(self, 0)
    if file is not None:
        print("\n".join(lines), file=file)
    return count
