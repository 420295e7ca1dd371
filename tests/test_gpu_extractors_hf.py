"""HIP feature extractors against tests/golden/G12_extractors_hf.npz — values produced by Hugging Face
transformers' EfficientNet-B0 / ResNet-18 alone (tests/golden/make_golden_hf.py, tests/hf_pin.py), i.e. an
implementation independent of both this repository's kernels and its oracle. When transformers is importable on
the GPU box the same comparison is also made against the live HF model.

Reference boundary this pins: `create_feature_extractor` -> timm `tf_efficientnet_b0`
(`/root/reference/model/feature_extractors.py:39-43`), FiLM-tagged layers `model/film.py:41-48`."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import orbit_dataset_amd  # noqa: E402,F401
import hf_pin  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G12_extractors_hf.npz")
TOL = 2e-5  # fp32 features of magnitude O(1): MFMA / DPP summation orders differ from the CPU's


@pytest.fixture(scope="module")
def g12():
    return dict(np.load(GOLD))


def _native(name, learn=False):
    fe, _ = create_feature_extractor(name, True, False, learn)
    synthetic.init_parameters_(fe)
    return fe.cuda()


@pytest.mark.parametrize("case,name,size,n,variant", hf_pin.CASES)
def test_hip_extractor_matches_transformers_fixture(device, g12, case, name, size, n, variant):
    fe = _native(name).eval()
    x = hf_pin.fixture_inputs(g12)[size][:n]
    with torch.no_grad():
        got = fe(x.to(device)).cpu()
    want = torch.from_numpy(g12[case + "_feats"])
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err < TOL, "%s: HIP vs transformers max |d feature| %g" % (case, err)


@pytest.mark.parametrize("case,name,size,n", hf_pin.TRAIN_CASES)
def test_hip_train_mode_batchnorm_matches_transformers_fixture(device, g12, case, name, size, n):
    """batch-statistics BatchNorm + running-statistics update of the native training runtime."""
    fe = _native(name, learn=True).train()
    x = hf_pin.fixture_inputs(g12)[size][:n]
    with torch.no_grad():
        got = fe(x.to(device)).cpu()
    want = torch.from_numpy(g12[case + "_train_feats"])
    assert (got - want).abs().max().item() < 5e-5
    sd = {k: v.cpu() for k, v in fe.state_dict().items()}
    for bn in hf_pin.TRAIN_STAT_KEYS[name]:
        for leaf in ("running_mean", "running_var"):
            w = torch.from_numpy(g12["%s_train_%s.%s" % (case, bn, leaf)])
            assert (sd[bn + "." + leaf] - w).abs().max().item() < 2e-5 * max(1.0, w.abs().max().item()), (bn, leaf)
        assert int(sd[bn + ".num_batches_tracked"]) == 1


@pytest.mark.parametrize("name,size,n", [("efficientnet_b0", 224, 6), ("resnet18", 224, 6), ("resnet18", 84, 16)])
def test_hip_extractor_matches_live_transformers(device, name, size, n):
    """Fresh inputs (not the fixture's) through the live HF model, when transformers is installed on the box."""
    pytest.importorskip("transformers")
    fe = _native(name).eval()
    x = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(size + n))
    model = hf_pin.hf_model(name, {k: v.cpu() for k, v in fe.state_dict().items()}).eval()
    with torch.no_grad():
        want = hf_pin.hf_features(model, x)
        got = fe(x.to(device)).cpu()
    assert (got - want).abs().max().item() < TOL
