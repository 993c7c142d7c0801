"""Host-side helpers of the hot path (srl/rl/functions.py): value rescaling (:10-17), per-actor
epsilon / beta / discount tables (:113-154), tie-breaking argmax (:157-172), invalid-action fancy index
(:217-229).  Pinned against the reference by tests/golden/functions.npz."""
import random
from typing import List, Union

import numpy as np


def rescaling(x, eps=0.001):
    return np.sign(x) * (np.sqrt(np.abs(x) + 1.0) - 1.0) + eps * x


def inverse_rescaling(x, eps=0.001):
    n = np.sqrt(1.0 + 4.0 * eps * (np.abs(x) + 1.0 + eps)) - 1.0
    n = n / (2.0 * eps)
    return np.sign(x) * ((n**2) - 1.0)


def sigmoid(x, a=1):
    return 1 / (1 + np.exp(-a * x))


def create_beta_list(policy_num: int, max_beta=0.3):
    assert policy_num > 0
    out = []
    for i in range(policy_num):
        if i == 0:
            b = 0
        elif i == policy_num - 1:
            b = max_beta
        else:
            b = 10 * (2 * i - (policy_num - 2)) / (policy_num - 2)
            b = max_beta * sigmoid(b)
        out.append(b)
    return out


def create_discount_list(policy_num: int, gamma0=0.9999, gamma1=0.997, gamma2=0.99):
    assert policy_num > 0
    out = []
    for i in range(policy_num):
        if i == 0:
            g = gamma0
        elif 1 <= i <= 6:
            g = gamma0 + (gamma1 - gamma0) * sigmoid(10 * ((2 * i - 6) / 6))
        elif i == 7:
            g = gamma1
        else:
            g = (policy_num - 9 - (i - 8)) * np.log(1 - gamma1) + (i - 8) * np.log(1 - gamma2)
            g = 1 - np.exp(g / (policy_num - 9))
        out.append(g)
    return out


def create_epsilon_list(policy_num: int, epsilon=0.4, alpha=8.0):
    assert policy_num > 0
    if policy_num == 1:
        return [epsilon / 4]
    return [epsilon ** (1 + (i / (policy_num - 1)) * alpha) for i in range(policy_num)]


def get_random_max_index(arr: Union[np.ndarray, List[float]], invalid_actions: List[int] = []) -> int:
    """argmax with uniformly random tie-breaking; invalid actions masked with -inf."""
    if len(arr) < 100:
        vals = arr.tolist() if isinstance(arr, np.ndarray) else list(arr)
        for a in invalid_actions:
            vals[a] = -np.inf
        best = max(vals)
        idx = [i for i, v in enumerate(vals) if v == best]
        return idx[0] if len(idx) == 1 else random.choice(idx)
    a = np.asarray(arr, dtype=float).copy()
    a[invalid_actions] = -np.inf
    return random.choice(np.where(a == a.max())[0].tolist())


def create_fancy_index_for_invalid_actions(idx_list: List[List[int]]):
    idx1 = [i for i, sub in enumerate(idx_list) for _ in sub]
    idx2 = [e for sub in idx_list for e in sub]
    return idx1, idx2


def random_choice_by_probs(probs, total=None):
    """Roulette selection with ONE `random.random()` draw (srl/rl/functions.py:183-194)."""
    if total is None:
        total = sum(probs)
    r = random.random() * total
    acc = 0
    for i, weight in enumerate(probs):
        acc += weight
        if r <= acc:
            return i
    raise ValueError(f"not coming. total: {total}, r: {r}, num: {acc}, probs: {probs}")


def calc_epsilon_greedy_probs(q, invalid_actions, epsilon, action_num):
    """epsilon-greedy as a probability vector: epsilon spread over the valid actions, the rest over the
    (possibly tied) maxima (srl/rl/functions.py:197-214)."""
    qv = np.array([(-np.inf if a in invalid_actions else v) for a, v in enumerate(q)])
    q_max = np.amax(qv, axis=0)
    n_max = np.count_nonzero(qv == q_max)
    n_valid = action_num - len(invalid_actions)
    probs = []
    for a in range(action_num):
        if a in invalid_actions:
            probs.append(0.0)
            continue
        p = epsilon / n_valid
        if qv[a] == q_max:
            p += (1 - epsilon) / n_max
        probs.append(p)
    return probs
