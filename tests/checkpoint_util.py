"""Shared by the checkpoint-compatibility tests: rebuild the reference-written checkpoint of
tests/golden/G13_checkpoint.npz from its recorded key list (values are a pure function of (seed, key), see
make_golden.py::g13_checkpoint) and write it to disk the way the reference does (torch.save of a state_dict)."""
import os

import numpy as np
import torch

from orbit_dataset_amd import synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G13_checkpoint.npz")
CASES = {"effnet_film": ("efficientnet_b0", True), "resnet_proto": ("resnet18", False)}


def load_gold():
    return dict(np.load(GOLD, allow_pickle=False))


def reference_layout_checkpoint(g, tag):
    """OrderedDict-free equivalent of what the reference saved: {key: tensor} in the reference's key order."""
    keys = [str(k) for k in g[tag + "_keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in g[tag + "_shapes"]]
    sub = ("feature_extractor.", "set_encoder.")
    probe = {k: torch.empty(s) for k, s in zip(keys, shapes)}

    class _Keys:  # just enough of a module for synthetic_state_dict's calibration overlay
        def state_dict(self):
            return probe
    extractor_sd = synthetic.synthetic_state_dict(_Keys())  # seed 1991 + BatchNorm calibration = the "pretrained" values
    ckpt = {}
    for k, s in zip(keys, shapes):
        if k.startswith("feature_extractor."):
            ckpt[k] = extractor_sd[k].clone()
        else:
            ckpt[k] = synthetic.synth_tensor(k, s, seed=7, film_strength=0.02)
    del sub
    return ckpt


def write_checkpoint(g, tag, path):
    torch.save(reference_layout_checkpoint(g, tag), path)
    return path
