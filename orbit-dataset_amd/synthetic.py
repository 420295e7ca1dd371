"""ORBIT-shaped synthetic tasks and deterministic parameters.

The ORBIT dataset (54-83 GB), the pretrained checkpoints and timm are all unavailable offline, so parity and
throughput are measured on synthetic tasks with the layout the reference's data pipeline emits
(reference data/datasets.py:584-597: `context_clips [N,T,3,H,W]` f32, `context_labels [N]` i64,
`target_clips`, `target_labels`) and on deterministically initialised parameters (same values on every
machine: each tensor is drawn from a torch CPU generator seeded by (seed, crc32(state_dict key))).

Frames emulate normalised pixels (reference data/datasets.py:82-83,430): frame = 0.5 * class template + unit
noise, so classes are separable but not trivially (argmax parity is a real test). `template="blobs"` draws LOW-FREQUENCY
class templates instead (7x7 colour grids, bilinearly enlarged) with a per-frame low-frequency distractor and pixel noise:
objects that differ in coarse colour layout, the family the meta-trained checkpoint (tools/meta_train.py) is trained and
evaluated on - a white-noise template carries no signal a convolutional extractor could be trained to keep. Parameters use He-normal
convolutions and NON-trivial BatchNorm statistics (gamma ~ U(.5,1.5), beta, running_mean ~ N(0,.1),
running_var ~ U(.5,1.5)) so that BN folding and FiLM are genuinely exercised; the last BatchNorm of every
residual branch is scaled down so activations keep O(1) magnitude through the depth of the network.
"""
import os
import zlib

import numpy as np
import torch

DEFAULT_SEED = 1991  # the reference's default --seed (utils/args.py:99)
_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
_CALIBRATION = {}
_SUBNETS = ("feature_extractor.", "set_encoder.")


def _calibration(name):
    """BatchNorm running statistics measured for the synthetic checkpoint `name` (a shipped data asset,
    produced by oracle/calibrate.py), or {} if the asset is absent."""
    if name not in _CALIBRATION:
        path = os.path.join(_ASSETS, "bn_calibration_%s.npz" % name)
        _CALIBRATION[name] = dict(np.load(path)) if os.path.exists(path) else {}
    return _CALIBRATION[name]


def _network_of(module):
    """Which synthetic checkpoint a module's keys belong to, judged from its state_dict keys."""
    keys = module if isinstance(module, (set, list, tuple)) else set(module.state_dict().keys())
    if "conv_stem.weight" in keys:
        return "efficientnet_b0"
    if "layer4.1.conv2.weight" in keys:
        return "resnet18"
    if "encoder.layer5.0.weight" in keys:
        return "set_encoder"
    return None


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def _randn(shape, g):
    return torch.randn(shape, generator=g, dtype=torch.float32)


def _rand(shape, g, lo, hi):
    return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo


def _is_branch_tail_bn(key):
    """last BatchNorm of a residual branch: resnet BasicBlock.bn2, EfficientNet InvertedResidual.bn3."""
    parts = key.split(".")
    if len(parts) < 2:
        return False
    in_resnet_layer = any(p.startswith("layer") and p[5:].isdigit() for p in parts[:-2])
    return (in_resnet_layer and parts[-2] == "bn2") or parts[-2] == "bn3"


def synth_tensor(key, shape, seed=DEFAULT_SEED, film_strength=0.1):
    """Deterministic value for the state_dict entry `key` of the given shape."""
    g = _gen(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return 0.1 * _randn(shape, g)
    if leaf == "running_var":
        return _rand(shape, g, 0.5, 1.5)
    if "regularizers" in key:  # FiLM regulariser r: the reference draws N(0, 0.001) (feature_adapters.py:50);
        return film_strength * _randn(shape, g)  # larger here so FiLM visibly modulates the features
    if len(shape) == 4:  # convolution [Cout, Cin/groups, KH, KW]
        fan_in = shape[1] * shape[2] * shape[3]
        gain = 1.0 if ".se." in key else 2.0 ** 0.5
        return (gain / fan_in ** 0.5) * _randn(shape, g)
    if len(shape) == 2:  # linear [out, in]
        return (1.0 / shape[1] ** 0.5) * _randn(shape, g)
    if leaf == "weight":  # BatchNorm / LayerNorm scale
        w = _rand(shape, g, 0.5, 1.5)
        return 0.25 * w if _is_branch_tail_bn(key) else w
    if leaf == "bias":
        return 0.1 * _randn(shape, g)
    return 0.1 * _randn(shape, g)


def synthetic_state_dict(module, seed=DEFAULT_SEED, prefix="", film_strength=0.1, use_calibration=True):
    """{key: tensor} for every entry of module.state_dict(), values determined by (seed, prefix + key).

    With `use_calibration` (and seed == DEFAULT_SEED, the seed the assets were measured for) the BatchNorm
    running statistics of a recognised network (or of recognised sub-networks `feature_extractor.` /
    `set_encoder.` of a recogniser) are overlaid from the shipped calibration asset."""
    def seed_key(k):
        # a sub-network gets the same values whether it is initialised alone or inside a recogniser
        for sub in _SUBNETS:
            if k.startswith(sub):
                return k[len(sub):]
        return k

    sd = {k: synth_tensor(prefix + seed_key(k), tuple(v.shape), seed, film_strength)
          for k, v in module.state_dict().items()}
    if use_calibration and seed == DEFAULT_SEED and prefix == "":
        for sub in ("",) + _SUBNETS:
            keys = {k[len(sub):] for k in sd if k.startswith(sub)}
            net = _network_of(keys)
            if net is None:
                continue
            for k, v in _calibration(net).items():
                if sub + k in sd and tuple(sd[sub + k].shape) == v.shape:
                    sd[sub + k] = torch.from_numpy(v.copy())
    return sd


def init_parameters_(module, seed=DEFAULT_SEED, prefix="", film_strength=0.1, use_calibration=True):
    """In-place deterministic initialisation of any module of this path (product or checker side)."""
    sd = synthetic_state_dict(module, seed, prefix, film_strength, use_calibration)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(sd[k].to(v.device))
    refresh = getattr(module, "refresh_initial_film_parameters", None)
    if refresh is not None:  # a recogniser with a FiLM generator: its snapshot of the extractor's BatchNorm follows
        refresh()
    return module


BLOBS = (0.9, 0.6, 0.5)  # template, per-frame distractor and pixel-noise amplitudes of the "blobs" family


def _lowfreq(n, H, W, g, device="cpu"):
    """n smooth random colour fields [n,3,H,W]: 7x7 grids of N(0,1) enlarged bilinearly (unit-ish variance)."""
    coarse = torch.randn(n, 3, 7, 7, generator=g, device=device)
    return torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False) * 1.4


def make_task(task_index=0, way=5, shots=5, frames_per_shot=8, num_query=200, frame_size=84, clip_length=1,
              seed=DEFAULT_SEED, device="cpu", label_values=None, dtype=torch.float32, template="noise"):
    """One synthetic task in the reference's task_dict layout.

    Support: way * shots * frames_per_shot frames grouped into clips of `clip_length` frames
    (N = way*shots*frames_per_shot / clip_length clips, labels shuffled as the reference shuffles train tasks,
    data/datasets.py:508-522). Query: `num_query` clips with labels uniform over the classes.
    `label_values` optionally maps class index -> label value (non-contiguous labels, e.g. (3, 7, 9)).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) + int(task_index))
    H = W = int(frame_size)
    T = int(clip_length)
    frames_per_class = shots * frames_per_shot
    assert frames_per_class % T == 0, "frames per class must be a multiple of clip_length"
    clips_per_class = frames_per_class // T
    N = way * clips_per_class
    if template == "blobs":
        a, b, c = BLOBS
        templates = _lowfreq(way, H, W, g)
        cls = torch.arange(way).repeat_interleave(clips_per_class)
        cls = cls[torch.randperm(N, generator=g)]
        context_clips = (a * templates[cls][:, None] + b * _lowfreq(N * T, H, W, g).view(N, T, 3, H, W)
                         + c * torch.randn(N, T, 3, H, W, generator=g))
        qcls = torch.randint(0, way, (num_query,), generator=g)
        target_clips = (a * templates[qcls][:, None] + b * _lowfreq(num_query * T, H, W, g).view(num_query, T, 3, H, W)
                        + c * torch.randn(num_query, T, 3, H, W, generator=g))
    else:
        templates = torch.randn(way, 3, H, W, generator=g)
        cls = torch.arange(way).repeat_interleave(clips_per_class)
        cls = cls[torch.randperm(N, generator=g)]
        context_clips = 0.5 * templates[cls][:, None] + torch.randn(N, T, 3, H, W, generator=g)
        qcls = torch.randint(0, way, (num_query,), generator=g)
        target_clips = 0.5 * templates[qcls][:, None] + torch.randn(num_query, T, 3, H, W, generator=g)
    values = torch.arange(way) if label_values is None else torch.as_tensor(label_values, dtype=torch.long)
    task = {
        "context_clips": context_clips.to(dtype),
        "context_labels": values[cls].long(),
        "target_clips": target_clips.to(dtype),
        "target_labels": values[qcls].long(),
        "object_list": ["object_%d" % int(v) for v in values],
    }
    if device != "cpu":
        task = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}
    return task


def make_task_on_device(task_index, way, shots, frames_per_shot, num_query, frame_size, clip_length, device,
                        seed=DEFAULT_SEED, template="noise"):
    """Same distribution as make_task but drawn directly in HBM (for throughput runs: no 100+ MB host copy).
    Values differ from make_task's (different generator); use make_task when CPU/GPU parity is checked."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + int(task_index))
    H = W = int(frame_size)
    T = int(clip_length)
    clips_per_class = shots * frames_per_shot // T
    N = way * clips_per_class
    blobs = template == "blobs"
    a, b, c = BLOBS if blobs else (0.5, 0.0, 1.0)
    templates = _lowfreq(way, H, W, g, device) if blobs else torch.randn(way, 3, H, W, generator=g, device=device)
    cls = torch.arange(way, device=device).repeat_interleave(clips_per_class)
    cls = cls[torch.randperm(N, generator=g, device=device)]
    context_clips = torch.randn(N, T, 3, H, W, generator=g, device=device)
    if blobs:
        context_clips.mul_(c).add_(_lowfreq(N * T, H, W, g, device).view(N, T, 3, H, W), alpha=b)
    context_clips.add_(templates[cls][:, None], alpha=a)
    qcls = torch.randint(0, way, (num_query,), generator=g, device=device)
    target_clips = torch.randn(num_query, T, 3, H, W, generator=g, device=device)
    if blobs:
        target_clips.mul_(c).add_(_lowfreq(num_query * T, H, W, g, device).view(num_query, T, 3, H, W), alpha=b)
    target_clips.add_(templates[qcls][:, None], alpha=a)
    return {"context_clips": context_clips, "context_labels": cls.long(), "target_clips": target_clips,
            "target_labels": qcls.long(), "object_list": ["object_%d" % i for i in range(way)]}
