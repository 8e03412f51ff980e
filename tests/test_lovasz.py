"""CPU: known-answer tests for LovaszSoftmax (temporal_latticenet_amd/lovasz.py), the loss train_ln.py:17, 119, 214 takes
from the un-vendored `latticenet_py.lattice.lovasz_loss` (called on log-softmax predictions [N, C] and int targets [N]).
Nothing under /root/reference pins its values, so the checks come from the definition (Berman, Triki, Blaschko, CVPR 2018):
the per-class loss is the Lovasz extension of the Jaccard set loss  Delta_c(M) = |M| / |{y = c} u M|  (M = mispredicted
points of class c), evaluated at the error vector m_i = |[y_i = c] - p_i(c)|, averaged over the classes present."""
import itertools

import numpy as np
import pytest
import torch

from temporal_latticenet_amd.lovasz import LovaszSoftmax


def _jaccard_set_loss(fg, mis):
    """Delta_c(M) from set operations alone: 1 - |F n P| / |F u P| with P the predicted foreground after flipping M"""
    fg = set(fg)
    mis = set(mis)
    pred = (fg - mis) | (mis - fg)
    union = fg | pred
    return 0.0 if not union else 1.0 - len(fg & pred) / len(union)


def _lovasz_extension_bruteforce(errors, fg):
    """sum_i m_pi(i) * [Delta({pi_1..pi_i}) - Delta({pi_1..pi_{i-1}})], pi sorting m in decreasing order — straight from the
    definition of the Lovasz extension of a set function, every Delta from set operations"""
    order = sorted(range(len(errors)), key=lambda i: -errors[i])
    fgset = [i for i in range(len(errors)) if fg[i]]
    total, prev, chosen = 0.0, 0.0, []
    for i in order:
        chosen.append(i)
        cur = _jaccard_set_loss(fgset, chosen)
        total += errors[i] * (cur - prev)
        prev = cur
    return total


def _logp(p):
    return torch.log(torch.as_tensor(p, dtype=torch.float64).clamp_min(1e-300)).float()


def test_perfect_prediction_costs_nothing():
    target = torch.tensor([0, 1, 2, 2, 1, 0, 3])
    probs = torch.nn.functional.one_hot(target, 4).float()
    loss = LovaszSoftmax()(torch.log(probs), target)               # log(0) = -inf -> exp = 0: as a hard log-softmax
    assert float(loss) == 0.0
    assert float(LovaszSoftmax(ignore_index=0)(torch.log(probs), target)) == 0.0


@pytest.mark.parametrize("seed", range(6))
def test_hard_predictions_give_one_minus_mean_iou(seed):
    """on the vertices of the hypercube the Lovasz extension IS the set function: a hard (one-hot) prediction costs
    mean over present classes of (1 - IoU_c) — independent of any sorting or tie order"""
    rng = np.random.default_rng(seed)
    n, c = 8, 3
    target = rng.integers(0, c, n)
    pred = rng.integers(0, c, n)
    probs = np.eye(c)[pred]
    want = []
    for k in range(c):
        f, p = target == k, pred == k
        if f.sum() == 0:
            continue                                                # classes="present"
        want.append(1.0 - (f & p).sum() / (f | p).sum())
    got = float(LovaszSoftmax()(_logp(probs), torch.from_numpy(target)))
    assert abs(got - float(np.mean(want))) < 1e-6


@pytest.mark.parametrize("seed", range(6))
def test_soft_predictions_match_the_bruteforce_extension(seed):
    rng = np.random.default_rng(100 + seed)
    n, c = 7, 3
    target = rng.integers(0, c, n)
    logits = rng.standard_normal((n, c))
    probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    want = []
    for k in range(c):
        fg = (target == k)
        if fg.sum() == 0:
            continue
        errors = np.abs(fg.astype(np.float64) - probs[:, k])
        want.append(_lovasz_extension_bruteforce(list(errors), list(fg)))
    got = float(LovaszSoftmax()(torch.log_softmax(torch.from_numpy(logits).float(), 1), torch.from_numpy(target)))
    assert abs(got - float(np.mean(want))) < 1e-6
    # the extension is convex and its value lies between 0 and 1
    assert 0.0 <= got <= 1.0


def test_the_set_function_check_itself():
    """the brute-force helper on all subsets of a tiny problem: Delta(M) = |M| / |F u M| (the paper's eq. for the Jaccard loss)"""
    fg = [0, 2]
    for r in range(5):
        for mis in itertools.combinations(range(4), r):
            union = set(fg) | set(mis)
            assert abs(_jaccard_set_loss(fg, mis) - (len(mis) / len(union) if union else 0.0)) < 1e-12


def test_ignore_index_removes_points_and_nothing_else():
    rng = np.random.default_rng(5)
    n, c = 40, 5
    target = torch.from_numpy(rng.integers(0, c, n))
    lp = torch.log_softmax(torch.from_numpy(rng.standard_normal((n, c))).float(), 1)
    keep = target != 0
    with_ignore = LovaszSoftmax(ignore_index=0)(lp, target)
    dropped = LovaszSoftmax()(lp[keep], target[keep])
    assert float(with_ignore) == float(dropped)
    # what an ignored point predicts does not matter
    lp2 = lp.clone()
    lp2[~keep] = torch.log_softmax(torch.from_numpy(rng.standard_normal((int((~keep).sum()), c))).float(), 1)
    assert float(LovaszSoftmax(ignore_index=0)(lp2, target)) == float(with_ignore)
    # everything ignored: zero, and differentiable (train_ln.py:214-216 calls backward on the sum with the NLL term)
    lp3 = lp.clone().requires_grad_(True)
    z = LovaszSoftmax(ignore_index=0)(lp3, torch.zeros(n, dtype=torch.long))
    z.backward()
    assert float(z.detach()) == 0.0 and float(lp3.grad.abs().max()) == 0.0


def test_gradient_is_the_lovasz_gradient_of_the_sorted_errors():
    """d loss / d p_i(c) = -/+ g_rank(i): the loss is piecewise linear in the errors"""
    rng = np.random.default_rng(9)
    n, c = 9, 2
    target = torch.from_numpy(rng.integers(0, c, n))
    logits = torch.from_numpy(rng.standard_normal((n, c))).double().requires_grad_(True)
    loss = LovaszSoftmax()(torch.log_softmax(logits, 1), target)
    loss.backward()
    g = logits.grad.clone()
    eps = 1e-6
    num = torch.zeros_like(g)
    for i in range(n):
        for k in range(c):
            d = torch.zeros_like(logits)
            d[i, k] = eps
            lp = LovaszSoftmax()(torch.log_softmax(logits.detach() + d, 1), target)
            lm = LovaszSoftmax()(torch.log_softmax(logits.detach() - d, 1), target)
            num[i, k] = (lp - lm) / (2 * eps)
    assert float((g - num).abs().max()) < 1e-6
