"""§8(f3) checkpoint compatibility, GPU half: a reference-layout checkpoint written to disk, loaded the way
single-step-learner.py:300-305 loads it (`model.load_state_dict(torch.load(path))`), then personalise() + predict() on
the HIP path must give the logits the REFERENCE computed after loading the same checkpoint (G13_checkpoint.npz) —
including the FiLM generator's gamma0/beta0 snapshot, which lives outside the state_dict (feature_adapters.py:55-58)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import orbit_dataset_amd  # noqa: E402,F401
import checkpoint_util as cu  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402


@pytest.fixture(scope="module")
def g13():
    return cu.load_gold()


def _task(g13, device):
    return tuple(torch.from_numpy(g13[k]).to(device) for k in ("context_clips", "context_labels", "target_clips"))


@pytest.mark.parametrize("tag", sorted(cu.CASES))
@pytest.mark.parametrize("load_on", ["host_then_move", "device"])
def test_loaded_checkpoint_reproduces_reference_logits(device, g13, tag, load_on, tmp_path):
    fe_name, adapt = cu.CASES[tag]
    path = cu.write_checkpoint(g13, tag, str(tmp_path / "checkpoint.pt"))
    model = SingleStepFewShotRecogniser(fe_name, adapt, "proto", 1, 4, False, 16, 1.0)
    model._set_device(device)
    if load_on == "device":  # reference order: init_model() sends to the device, then load_state_dict (learner :300-302)
        model._send_to_device()
        model.load_state_dict(torch.load(path))
    else:
        model.load_state_dict(torch.load(path, map_location="cpu"))
        model._send_to_device()
    model.set_test_mode(True)
    ctx, lab, tgt = _task(g13, device)
    with torch.no_grad():
        model.personalise(ctx, lab)
        got = model.predict(tgt).cpu()
        if adapt:
            film_bn1 = model.film_dict["bn1.weight"].cpu()
    want = torch.from_numpy(g13[tag + "_logits"])
    assert (got - want).abs().max().item() < 1e-3
    assert torch.equal(got.argmax(1), want.argmax(1))
    if adapt:
        assert torch.allclose(film_bn1, torch.from_numpy(g13[tag + "_film_bn1_weight"]), atol=1e-5)
        assert float(film_bn1.abs().min()) > 0  # an all-zero snapshot would zero every FiLM gamma
    model._reset()


def test_learner_model_path_and_saved_checkpoint(device, g13, tmp_path):
    """learner.py --model_path on a reference-layout file; --mode train_test writes a checkpoint it can re-load."""
    from orbit_dataset_amd import learner
    path = cu.write_checkpoint(g13, "effnet_film", str(tmp_path / "checkpoint.pt"))
    base = ["--feature_extractor", "efficientnet_b0", "--adapt_features", "--frame_size", "64", "--way", "3", "--shots", "1",
            "--frames_per_shot", "4", "--num_query_videos", "2", "--frames_per_video", "6", "--num_test_tasks", "2",
            "--batch_size", "8"]
    stats = learner.main(base + ["--mode", "test", "--model_path", path])
    assert 0.0 <= stats["test"]["frame_acc"][0] <= 1.0
    out = str(tmp_path / "trained.pt")
    stats = learner.main(base + ["--mode", "train_test", "--model_path", path, "--with_lite", "--num_lite_samples", "4",
                                 "--num_train_tasks", "2", "--tasks_per_batch", "2", "--save_model_path", out,
                                 "--learning_rate", "1e-3"])
    assert os.path.exists(out)
    saved = torch.load(out)
    ckpt = torch.load(path)
    assert sorted(saved) == sorted(ckpt)  # same keys as the reference's files (order is the module tree's)
    # the frozen extractor is unchanged, the set encoder / FiLM generator moved
    assert torch.equal(saved["feature_extractor.conv_stem.weight"].cpu(), ckpt["feature_extractor.conv_stem.weight"])
    assert not torch.equal(saved["film_generator.regularizers.0"].cpu(), ckpt["film_generator.regularizers.0"])
    stats2 = learner.main(base + ["--mode", "test", "--model_path", out])
    assert abs(stats2["test"]["frame_acc"][0] - stats["test"]["frame_acc"][0]) < 1e-6
