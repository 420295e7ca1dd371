"""The accuracy half of BASELINE's metric: the LITE meta-training loop of the reference
(single-step-learner.py:136-194) run through the native forward / backward kernels CONVERGES - the loss falls and the
held-out frame accuracy rises well above chance - and the committed meta-trained checkpoint gives the same logits on the HIP
path as on the CPU oracle loading the same file (single-step-learner.py:300-305: the reference tests on trained weights)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from oracle.recogniser import OracleRecogniser  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(ROOT, "orbit-dataset_amd", "assets", "meta_trained_efficientnet_b0_224.npz")


def test_lite_meta_training_converges(device, tmp_path):
    out = str(tmp_path / "ckpt.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "meta_train.py"), "--steps", "160", "--frame_size", "96",
                        "--eval_tasks", "4", "--out", out], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    log = json.load(open(out[:-4] + ".json"))
    assert log["loss_last20"] < 0.25 * log["loss_first20"], (log["loss_first20"], log["loss_last20"])
    assert log["heldout_acc_after"] >= 0.8 and log["heldout_acc_after"] >= log["heldout_acc_before"] + 0.2, log
    assert os.path.getsize(out) < 12e6


def _load(path):
    return {k: torch.from_numpy(v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in np.load(path).items()}


def test_meta_trained_checkpoint_parity_and_accuracy(device):
    """The committed checkpoint (tools/meta_train.py, 300 LITE tasks at 224x224): HIP logits == oracle logits (1e-3,
    identical argmax) on a 100 + 60-frame task of the family it was trained on, and both classify it well above chance."""
    assert os.path.exists(CKPT), "run tools/meta_train.py"
    sd = _load(CKPT)
    model = SingleStepFewShotRecogniser("efficientnet_b0", False, "proto", 1, 256, False, 16, 1.0)
    model.load_state_dict(sd)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(7, way=5, shots=1, frames_per_shot=20, num_query=60, frame_size=224, template="blobs")
    with torch.no_grad():
        model.personalise(task["context_clips"].cuda(), task["context_labels"].cuda())
        got = model.predict(task["target_clips"].cuda()).cpu()
    ref = OracleRecogniser("efficientnet_b0", False, "proto", 1, 256)
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v for k, v in sd.items() if k.startswith("feature_extractor.")})
    ref.personalise(task["context_clips"], task["context_labels"])
    want = ref.predict(task["target_clips"])
    err = (got - want).abs().max().item()
    assert err <= 1e-3, "max |dlogit| %g (logit scale %g): north_star's bar is 1e-3 ABSOLUTE" % (err, want.abs().max().item())
    assert torch.equal(got.argmax(1), want.argmax(1))
    acc = (got.argmax(1) == task["target_labels"]).float().mean().item()
    assert acc >= 0.6, acc
