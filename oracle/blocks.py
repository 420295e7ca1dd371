"""ORACLE (test infrastructure) — PyTorch-CPU restatement of the model blocks downstream of the extractor.

Each function/class cites the reference lines it restates. Pinned against the imported reference by the
golden fixtures G1-G4, G7 (tests/golden/, tests/test_oracle_golden.py).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- data/utils.py --------------------------------------------------------------------------------------
def get_batch_indices(index, last_element, batch_size):
    """reference data/utils.py:49-54"""
    start = index * batch_size
    end = start + batch_size
    return start, (last_element if end > last_element else end)


def attach_frame_history(frames, history_length):
    """reference data/utils.py:8-28 — clip f = frames [f-L+1 .. f], indices below 0 clamped to frame 0."""
    L, n = history_length, frames.shape[0]
    out = []
    for f in range(n):
        out.append(torch.stack([frames[max(f - (L - 1) + j, 0)] for j in range(L)], dim=0))
    return torch.stack(out, dim=0)


# ---- model/poolers.py -------------------------------------------------------------------------------------
def mean_pool(x, T):
    """reference model/poolers.py:13-16"""
    return x.reshape(-1, T, x.shape[-1]).mean(dim=1)


# ---- model/classifier_heads.py ------------------------------------------------------------------------------
def proto_configure(features, labels, distance_fn="euclidean"):
    """reference model/classifier_heads.py:94-119 (_build_class_reps) + :232-263 (configure).
    Returns (class_ids ascending, W [C,D], b [C] or None)."""
    class_ids = sorted(set(int(v) for v in labels.tolist()))
    W, b = [], []
    for c in class_ids:
        mu = features[labels == c].mean(dim=0, keepdim=True)      # :116-119 mean pooling of the class rows
        W.append(2 * mu)                                             # :253
        if distance_fn == "euclidean":
            b.append(-(mu @ mu.t()).reshape(1))                      # :255
    W = torch.cat(W, dim=0)
    return class_ids, W, (torch.cat(b) if distance_fn == "euclidean" else None)


def proto_predict(features, W, b, logit_scale=1.0, distance_fn="euclidean"):
    """reference model/classifier_heads.py:202-219"""
    if W is None or (distance_fn == "euclidean" and b is None):
        raise AttributeError("Weight and/or bias not set - is model personalised?")
    if distance_fn == "euclidean":
        return logit_scale * (features @ W.t() + b)                 # :213 F.linear
    # :215-217 — cosine similarity of every (query, class) pair along the feature axis, eps 1e-8
    q = features[:, :, None].expand(-1, -1, W.shape[0])
    w = W.t()[None, :, :].expand(features.shape[0], -1, -1)
    return logit_scale * F.cosine_similarity(q, w, dim=1)


def frame_accuracy(logits, labels):
    """reference utils/eval_metrics.py:27-36"""
    return (logits.argmax(dim=-1) == labels).float().mean().item()


# ---- model/set_encoders.py -----------------------------------------------------------------------------------
class SimplePrePoolNet(nn.Module):
    """reference model/set_encoders.py:81-120"""

    def __init__(self):
        super().__init__()
        for i in range(5):
            setattr(self, f"layer{i + 1}", nn.Sequential(
                nn.Conv2d(3 if i == 0 else 64, 64, kernel_size=3, stride=1, padding=1),
                nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=False)))

    def forward(self, x):
        for i in range(5):
            x = getattr(self, f"layer{i + 1}")(x)
        return x.mean((2, 3))                                        # AdaptiveAvgPool2d((1,1)) + view


class SetEncoder(nn.Module):
    """reference model/set_encoders.py:34-79"""
    output_size = 64

    def __init__(self):
        super().__init__()
        self.encoder = SimplePrePoolNet()

    def forward(self, x):
        return self.encoder(x.flatten(end_dim=1) if x.dim() == 5 else x)

    @staticmethod
    def aggregate(x, aggregation="mean"):
        if not isinstance(x, torch.Tensor):
            x = torch.cat(x, dim=0)
        if aggregation == "mean":
            return x.mean(dim=0, keepdim=True)
        if aggregation == "none":
            return x
        raise ValueError(f"Aggregation method {aggregation} not valid!")


# ---- model/mlps.py + model/feature_adapters.py ---------------------------------------------------------------
class DenseBlock(nn.Module):
    """reference model/mlps.py:52-63"""

    def __init__(self, in_size, hidden_size, out_size):
        super().__init__()
        self.block = nn.Sequential(nn.Linear(in_size, hidden_size), nn.LayerNorm(hidden_size), nn.ReLU(),
                                   nn.Linear(hidden_size, out_size))

    def forward(self, x):
        return self.block(x)


class FilmParameterGenerator(nn.Module):
    """reference model/feature_adapters.py:36-78"""

    def __init__(self, sizes, initial, pooled_size=64, hidden_size=64):
        super().__init__()
        self.initial_film_parameters = initial
        self.film_parameter_names = sorted(initial.keys())            # :43-44
        self.generators = nn.ModuleList(DenseBlock(pooled_size, hidden_size, sizes[n])
                                        for n in self.film_parameter_names)
        self.regularizers = nn.ParameterList(nn.Parameter(torch.randn(sizes[n]) * 0.001)
                                             for n in self.film_parameter_names)
        self.l2_term = 0.0

    def regularization_term(self):
        return self.l2_term

    def forward(self, z):
        film, l2 = {}, 0.0
        for i, name in enumerate(self.film_parameter_names):
            g = self.generators[i](z).squeeze() * self.regularizers[i]
            if "weight" in name:
                film[name] = self.initial_film_parameters[name] * (g + 1.0)     # :70-71
            elif "bias" in name:
                film[name] = self.initial_film_parameters[name] + g             # :73-74
            l2 = l2 + (self.regularizers[i] ** 2).sum()                         # :76
        self.l2_term = l2
        return film


# ---- model/mlps.py:33-50 + model/classifier_heads.py:121-180 (Versa) and :265-368 (Mahalanobis) -------------------------
class DenseResidualBlock(nn.Module):
    """reference model/mlps.py:33-50"""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.linear1 = nn.Linear(in_size, out_size)
        self.linear2 = nn.Linear(out_size, out_size)
        self.linear3 = nn.Linear(out_size, out_size)
        self.elu = nn.ELU()

    def forward(self, x):
        out = self.linear3(self.elu(self.linear2(self.elu(self.linear1(x)))))
        return out + x if x.shape[-1] == out.shape[-1] else out


def _class_rows(features, labels):
    ids = torch.unique(labels)                                   # ascending = column order (:137-139, :292)
    return ids, [features[labels == c] for c in ids]


def versa_configure(features, labels, weight_processor, bias_processor):
    """reference classifier_heads.py:158-180 -> (class ids, weight [C, D], bias [C])"""
    ids, rows = _class_rows(features, labels)
    nus = [r.mean(dim=0, keepdim=True) for r in rows]            # _mean_pooling, :115-119
    weight = torch.cat([weight_processor(nu) for nu in nus], dim=0)
    bias = torch.cat([bias_processor(nu) for nu in nus], dim=1).reshape(len(ids))
    return ids, weight.detach(), bias.detach()


def estimate_cov(examples):
    """reference classifier_heads.py:349-368 (_estimate_cov), including its single-example branch, which centres the one
    example by its own mean over the features and returns a SCALAR that later broadcasts over the matrix"""
    if examples.size(0) > 1:
        return torch.cov(examples.t(), correction=1)
    factor = 1.0 / (examples.size(1) - 1)
    examples = examples - torch.mean(examples, dim=1, keepdim=True)
    return factor * examples.matmul(examples.t()).squeeze()


def mahalanobis_configure(features, labels):
    """reference classifier_heads.py:284-327 -> (ids, means [C, D], precisions [C, D, D], task_mean, task_precision)"""
    D = features.size(1)
    eye = torch.eye(D, dtype=features.dtype)
    task_cov = estimate_cov(features)
    task_precision = torch.inverse(task_cov + eye)
    ids, rows = _class_rows(features, labels)
    means, precisions = [], []
    for r in rows:
        means.append(r.mean(dim=0))
        lam = r.size(0) / (r.size(0) + 1)
        precisions.append(torch.inverse(lam * estimate_cov(r) + (1 - lam) * task_cov + eye))
    return ids, torch.stack(means), torch.stack(precisions), features.mean(dim=0), task_precision


def mahalanobis_predict(features, means, precisions, logit_scale=1.0):
    """reference classifier_heads.py:329-347"""
    diff = means[:, None, :] - features[None, :, :]              # [C, M, D]
    first_half = torch.matmul(diff, precisions)
    return logit_scale * (-(first_half * diff).sum(dim=2).transpose(1, 0))
