"""The HIP path against the golden vectors captured from the reference's own modules (tests/golden/*.npz):
the committed fixtures are what the product must reproduce on the GPU box, where /root/reference does not exist."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.data.utils import attach_frame_history, get_batch_indices  # noqa: E402
from orbit_dataset_amd.model.classifier_heads import PrototypicalClassifier  # noqa: E402
from orbit_dataset_amd.model.feature_adapters import FilmParameterGenerator  # noqa: E402
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402
from orbit_dataset_amd.model.film import get_film_parameter_sizes, get_film_parameters  # noqa: E402
from orbit_dataset_amd.model.poolers import MeanPooler  # noqa: E402
from orbit_dataset_amd.model.set_encoders import SetEncoder  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v)
            for k, v in np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False).items()}


@pytest.mark.parametrize("case", ["w5_d512", "w10_d96", "noncontig_d96", "oneshot_d64", "w5_d1280", "w10_d1280"])
@pytest.mark.parametrize("dist", ["euclidean", "cosine"])
@pytest.mark.parametrize("scale", [1, 32])
def test_G1_head(device, case, dist, scale):
    g = gold("G1_head")
    key = "%s_%s_s%d" % (case, dist, scale)
    head = PrototypicalClassifier(float(scale), dist)
    head.configure(g[case + "_feats"].to(device), g[case + "_labels"].to(device))
    logits = head.predict(g[case + "_q"].to(device)).cpu()
    want = g[key + "_logits"]
    assert torch.allclose(head.weight.cpu(), g[key + "_W"], atol=1e-6)
    if dist == "euclidean":
        assert torch.allclose(head.bias.cpu(), g[key + "_b"], atol=1e-4, rtol=1e-6)
    assert (logits - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    assert torch.equal(logits.argmax(1), want.argmax(1))
    if dist == "cosine":
        assert torch.all(logits[0] == 0)


def test_G2_pooler_G7_utils(device):
    g = gold("G2_pooler")
    for T in (1, 3, 8):
        assert torch.allclose(MeanPooler(T)(g["x"].to(device)).cpu(), g["T%d" % T], atol=1e-6)
    u = gold("G7_utils")
    assert torch.equal(attach_frame_history(u["frames"], 1), u["hist1"])
    assert torch.equal(attach_frame_history(u["frames"].to(device), 3).cpu(), u["hist3"])
    assert [list(get_batch_indices(i, 10, 4)) for i in range(3)] == u["batch_10_4"].tolist()


def test_G3_set_encoder(device):
    g = gold("G3_set_encoder")
    enc = SetEncoder()
    synthetic.init_parameters_(enc)
    enc = enc.cuda().eval()
    reps = enc(g["x"].to(device))
    assert (reps.cpu() - g["reps"]).abs().max().item() < 2e-5
    assert (enc.aggregate(reps).cpu() - g["mean"]).abs().max().item() < 2e-5
    assert (enc(g["x"][:, :, :, :32, :32].contiguous().to(device)).cpu() - g["reps32"]).abs().max().item() < 2e-5


def test_G4_film_generator(device):
    g = gold("G4_film_generator")
    fe, names = create_feature_extractor("efficientnet_b0", True, True, False)
    synthetic.init_parameters_(fe)
    gen = FilmParameterGenerator(get_film_parameter_sizes(names, fe), get_film_parameters(names, fe), 64, 64,
                                 slot_names=[n for n, _ in fe.film_slot_modules()])
    synthetic.init_parameters_(gen, prefix="film_generator.")
    gen = gen.cuda()
    assert gen.film_parameter_names == [str(n) for n in g["names"]]
    film = gen(g["z"].to(device))
    for i, n in enumerate(gen.film_parameter_names):
        assert (film[n].cpu() - g["film_%03d" % i]).abs().max().item() < 1e-5, n
    assert abs(float(gen.regularization_term()) - float(g["l2_term"])) < 1e-5 * float(g["l2_term"])


def native(adapt, classifier, clip_length, batch_size, num_lite=16, scale=1.0):
    m = SingleStepFewShotRecogniser("resnet18", adapt, classifier, clip_length, batch_size, False, num_lite, scale)
    synthetic.init_parameters_(m)
    if adapt:
        m.film_generator.initial_film_parameters = get_film_parameters(m.film_parameter_names, m.feature_extractor)
    m._set_device("cuda:0")
    m._send_to_device()
    return m


@pytest.mark.parametrize("tag,adapt,classifier,scale", [("proto", False, "proto", 1.0),
                                                        ("cosine", False, "proto_cosine", 32.0),
                                                        ("film", True, "proto", 1.0)])
def test_G5_recogniser(device, tag, adapt, classifier, scale):
    g = gold("G5_recogniser")
    m = native(adapt, classifier, 1, 4, scale=scale)
    m.set_test_mode(True)
    m.personalise(g["context_clips"], g["context_labels"].to(device))  # frames on the host, moved per mini-batch
    logits = m.predict(g["target_clips"].to(device)).cpu()
    want = g[tag + "_logits"]
    assert (logits - want).abs().max().item() < 1e-3
    assert torch.equal(logits.argmax(1), want.argmax(1))
    assert torch.allclose(m.classifier.weight.cpu(), g[tag + "_W"], atol=2e-5)
    if adapt:
        assert torch.allclose(m.film_dict["bn1.weight"].cpu(), g["film_film_bn1_weight"], atol=1e-5)
        assert abs(float(m.film_generator.regularization_term()) - float(g["film_l2"])) < 1e-5 * float(g["film_l2"])


def test_G5_clip_length_3(device):
    g = gold("G5_recogniser")
    m = native(False, "proto", 3, 2)
    m.set_test_mode(True)
    m.personalise(g["T3_context_clips"].to(device), g["T3_context_labels"].to(device))
    clips = attach_frame_history(g["T3_video"].to(device), m.clip_length)
    logits = m.predict(clips).cpu()
    assert m.classifier.class_ids.cpu().tolist() == [3, 7, 9]
    assert (logits - g["T3_logits"]).abs().max().item() < 1e-3
    assert torch.equal(logits.argmax(1), g["T3_logits"].argmax(1))


def test_G6_lite_forward(device):
    """Learner.train_task_with_lite's call order (single-step-learner.py:212-243) on the native model: per query
    batch {personalise_with_lite -> predict_a_batch -> scaled CE + 0.001 l2 -> _reset}; forward values only."""
    import torch.nn.functional as F
    g = gold("G6_lite")
    nl, tpb, bs = int(g["num_lite_samples"]), int(g["tasks_per_batch"]), int(g["batch_size"])
    m = native(True, "proto", 1, bs, num_lite=nl)
    m.set_test_mode(False)  # frozen extractor: BatchNorm stays in eval mode (few_shot_recognisers.py:176-183)
    ctx, lab = g["context_clips"].to(device), g["context_labels"].to(device)
    m._clear_caches()
    for b in range(2):
        np.random.seed(500 + b)
        m.personalise_with_lite(ctx, lab)
        logits = m.predict_a_batch(g["target_clips"][b * bs:(b + 1) * bs].to(device))
        want = g["logits_%d" % b]
        assert (logits.cpu() - want).abs().max().item() < 1e-3
        loss = len(lab) / (nl * tpb) * F.cross_entropy(logits, g["target_labels"][b * bs:(b + 1) * bs].to(device))
        loss = loss + 0.001 * m.film_generator.regularization_term()
        assert abs(float(loss) - float(g["loss_%d" % b])) < 1e-3
        m._reset()
