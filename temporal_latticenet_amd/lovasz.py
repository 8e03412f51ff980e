"""LovaszSoftmax(ignore_index) as imported at train_ln.py:17 / used at train_ln.py:119, 214 on log-softmax
predictions [N, C] and integer targets [N].  Written from the published algorithm (Berman, Triki, Blaschko,
"The Lovasz-Softmax loss", CVPR 2018, Alg. 1); training-side only, not on the forward hot path."""
import torch


def _lovasz_grad(gt_sorted):
    gts = gt_sorted.sum()
    intersection = gts - gt_sorted.cumsum(0)
    union = gts + (1.0 - gt_sorted).cumsum(0)
    jaccard = 1.0 - intersection / union
    if gt_sorted.numel() > 1:
        jaccard[1:] = jaccard[1:] - jaccard[:-1]
    return jaccard


class LovaszSoftmax(torch.nn.Module):
    def __init__(self, ignore_index=None, classes="present"):
        super().__init__()
        self.ignore_index = ignore_index
        self.classes = classes

    def forward(self, log_probs, target):
        probs = log_probs.exp()
        if self.ignore_index is not None:
            keep = target != self.ignore_index
            probs, target = probs[keep], target[keep]
        if probs.numel() == 0:
            return probs.sum() * 0.0
        losses = []
        for c in range(probs.shape[1]):
            fg = (target == c).to(probs.dtype)
            if self.classes == "present" and fg.sum() == 0:
                continue
            errors = (fg - probs[:, c]).abs()
            errors_sorted, perm = torch.sort(errors, 0, descending=True)
            losses.append(torch.dot(errors_sorted, _lovasz_grad(fg[perm])))
        if not losses:
            return probs.sum() * 0.0
        return torch.stack(losses).mean()
