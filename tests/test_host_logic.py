"""CPU tests of the host-side mirror: helpers, synthetic data determinism, parameter-tree compatibility with
the oracle's torchvision/timm-layout modules, FiLM bookkeeping, and that the product never imports oracle/."""
import os
import re

import numpy as np
import pytest
import torch

import orbit_dataset_amd  # noqa: F401
from oracle import blocks, extractors
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.data import utils as dutils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_batch_indices_matches_reference_probe():
    # SURVEY §8c G7 probe: 10 items, batch 4 -> (0,4),(4,8),(8,10)
    assert [dutils.get_batch_indices(i, 10, 4) for i in range(3)] == [(0, 4), (4, 8), (8, 10)]
    for n in (1, 7, 256, 257):
        for bs in (1, 3, 256):
            for i in range(-(-n // bs)):
                assert dutils.get_batch_indices(i, n, bs) == blocks.get_batch_indices(i, n, bs)


@pytest.mark.parametrize("L", [1, 2, 3, 8])
def test_attach_frame_history(L):
    frames = torch.arange(7 * 3 * 2 * 2, dtype=torch.float32).reshape(7, 3, 2, 2)
    got = dutils.attach_frame_history(frames, L)
    assert got.shape == (7, L, 3, 2, 2)
    assert torch.equal(got, blocks.attach_frame_history(frames, L))
    assert torch.equal(got[:, -1], frames)                     # last slot is the frame itself
    assert torch.equal(got[0], frames[:1].expand(L, -1, -1, -1))  # frame 0 padded with itself


def test_synthetic_task_layout_and_determinism():
    a = synthetic.make_task(3, way=5, shots=5, frames_per_shot=8, num_query=20, frame_size=16)
    b = synthetic.make_task(3, way=5, shots=5, frames_per_shot=8, num_query=20, frame_size=16)
    assert a["context_clips"].shape == (200, 1, 3, 16, 16) and a["context_clips"].dtype == torch.float32
    assert a["context_labels"].shape == (200,) and a["context_labels"].dtype == torch.int64
    assert a["target_clips"].shape == (20, 1, 3, 16, 16)
    assert all(torch.equal(a[k], b[k]) for k in ("context_clips", "context_labels", "target_clips", "target_labels"))
    assert torch.bincount(a["context_labels"]).tolist() == [40] * 5
    c = synthetic.make_task(4, way=5, shots=5, frames_per_shot=8, num_query=20, frame_size=16, clip_length=8)
    assert c["context_clips"].shape == (25, 8, 3, 16, 16)
    d = synthetic.make_task(0, way=3, shots=1, frames_per_shot=2, num_query=5, frame_size=8, label_values=(3, 7, 9))
    assert sorted(set(d["context_labels"].tolist())) == [3, 7, 9]


@pytest.mark.parametrize("name", ["resnet18", "efficientnet_b0"])
def test_parameter_tree_is_checkpoint_compatible(name):
    from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
    ref = extractors.create(name)
    fe, film_names = create_feature_extractor(name, True, True, False)
    a, b = ref.state_dict(), fe.state_dict()
    assert list(sorted(a)) == list(sorted(b))
    assert all(a[k].shape == b[k].shape for k in a)
    fe.load_state_dict(a, strict=True)
    assert not any(p.requires_grad for p in fe.parameters())         # learn_extractor=False freezes (:81-87)
    assert [n for n, _ in fe.film_slot_modules()] == ref.film_slot_names()
    expect = [s + sfx for s in ref.film_slot_names() for sfx in (".weight", ".bias")]
    assert film_names == expect
    synthetic.init_parameters_(fe)
    synthetic.init_parameters_(ref)
    assert all(torch.equal(fe.state_dict()[k], ref.state_dict()[k]) for k in a)


def test_recogniser_wiring_and_state_dict_names():
    from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser
    m = SingleStepFewShotRecogniser("efficientnet_b0", True, "proto", 1, 256, False, 16, 1.0)
    keys = set(m.state_dict().keys())
    # names the reference's checkpoints use (SURVEY §5 checkpoint row)
    for k in ("feature_extractor.conv_stem.weight", "set_encoder.encoder.layer1.0.weight",
              "film_generator.generators.0.block.0.weight", "film_generator.generators.33.block.3.bias",
              "film_generator.regularizers.33"):
        assert k in keys, k
    assert len(m.film_generator.film_parameter_names) == 34 and m.film_generator.film_size == 10240
    assert m.film_generator.film_parameter_names == sorted(m.film_parameter_names)
    assert m.clip_length == 1 and m.feature_extractor.output_size == 1280
    m2 = SingleStepFewShotRecogniser("resnet18", False, "proto_cosine", 8, 4, False, 16, 32.0)
    assert m2.classifier.distance_fn == "cosine" and m2.film_generator.regularization_term() == 0
    assert m2.film_generator(None) == {}
    m5 = SingleStepFewShotRecogniser("resnet18", True, "versa", 1, 4, False, 16)
    keys5 = set(m5.state_dict())
    assert {"classifier.weight_processor.linear1.weight", "classifier.bias_processor.linear3.bias"} <= keys5
    assert SingleStepFewShotRecogniser("resnet18", True, "mahalanobis", 1, 4, False, 16).classifier.means is None
    from orbit_dataset_amd.model.few_shot_recognisers import MultiStepFewShotRecogniser
    m6 = MultiStepFewShotRecogniser("resnet18", True, "linear", 1, 4, False)
    film_trainable = {n for n, p in m6.feature_extractor.named_parameters() if p.requires_grad}
    assert film_trainable == set(m6.film_parameter_names) and len(film_trainable) == 40  # unfreeze_film, :196-199
    with pytest.raises(ValueError):
        SingleStepFewShotRecogniser("resnet18", False, "nearest", 1, 4, False, 16)
    m3 = SingleStepFewShotRecogniser("resnet18", False, "proto", 1, 4, True, 16)
    m3.set_test_mode(False)
    # BatchNorm policy of the reference (few_shot_recognisers.py:176-183): everything eval(), the extractor train()
    # iff it is being learned and the model is not in test mode
    m3._set_batch_norm_state()
    assert m3.feature_extractor.training and not m3.classifier.training and not m3.training
    m3.set_test_mode(True)
    m3._set_batch_norm_state()
    assert not m3.feature_extractor.training
    m4 = SingleStepFewShotRecogniser("resnet18", True, "proto", 1, 4, False, 16)
    m4.set_test_mode(False)
    m4._set_batch_norm_state()
    assert not m4.feature_extractor.training and not m4.set_encoder.training


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "orbit-dataset_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
                assert "oracle/" not in src or f in ("synthetic.py",), os.path.join(dirpath, f)


def test_learner_cli_flags_match_reference_names():
    """The counterpart CLI accepts the reference's hot-path flag names (utils/args.py) and its training guard."""
    from orbit_dataset_amd.learner import build_parser, mean_ci, verify_args
    p = build_parser()
    a = p.parse_args(["--mode", "test", "--feature_extractor", "resnet18", "--classifier", "proto_cosine",
                      "--logit_scale", "32", "--clip_length", "8", "--frame_size", "84", "--batch_size", "64",
                      "--tasks_per_batch", "16", "--with_lite", "--num_lite_samples", "16", "--gpu", "0", "--seed", "7",
                      "--adapt_features", "--model_path", "x.pt"])
    assert (a.feature_extractor, a.classifier, a.logit_scale, a.clip_length, a.frame_size) == (
        "resnet18", "proto_cosine", 32.0, 8, 84)
    assert a.with_lite and a.adapt_features and not a.learn_extractor
    with pytest.raises(SystemExit):  # args.py:209-211: training needs something learnable
        verify_args(p.parse_args(["--mode", "train"]))
    m, ci = mean_ci([0.5, 0.7])
    assert abs(m - 0.6) < 1e-12 and abs(ci - 1.96 * 0.1 / 2 ** 0.5) < 1e-12


def test_mark_parameters_changed():
    """fused optimizers do not bump tensor versions, so the plans' (data_ptr, _version) stamps would miss their updates:
    the helper invalidates every native plan and the FiLM generator's cached upload"""
    import types
    from orbit_dataset_amd.optim import mark_parameters_changed
    p = torch.nn.Parameter(torch.randn(4))
    p.grad = torch.randn(4)
    v0 = p._version
    torch.optim.Adam([p], lr=0.1, fused=True).step()
    assert p._version == v0  # the hazard this guards against (if torch ever changes this, the hook is merely redundant)
    net = torch.nn.Module()
    net.__dict__["_plans"] = {(8, 8): types.SimpleNamespace(stamp=("x",))}
    gen = torch.nn.Module()
    gen.__dict__["_stamp"] = ("y",)
    root = torch.nn.Module()
    root.a, root.b = net, gen
    mark_parameters_changed(root)
    assert net.__dict__["_plans"][(8, 8)].stamp is None and gen.__dict__["_stamp"] is None



def test_unique_label_cache_is_weak_and_thread_safe():
    """the memo of `unique_labels` no longer pins label tensors alive and can be shared by threads."""
    import gc
    import threading
    from orbit_dataset_amd.model.classifier_heads import PrototypicalClassifier as P
    P._unique_cache.clear()
    lab = torch.tensor([7, 3, 3, 9, 7])
    ids = P.unique_labels(lab, "cpu")
    assert ids.tolist() == [3, 7, 9] and P.unique_labels(lab, "cpu") is ids and len(P._unique_cache) == 1
    lab += 0  # in-place op: new version -> new entry, not a stale hit
    assert P.unique_labels(lab, "cpu") is not ids
    del lab
    gc.collect()
    assert len(P._unique_cache) == 0  # entries die with their tensor
    errors = []

    def work(seed):
        try:
            g = torch.Generator().manual_seed(seed)
            for _ in range(200):
                t = torch.randint(0, 6, (12,), generator=g)
                assert P.unique_labels(t, "cpu").tolist() == sorted(set(t.tolist()))
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in threads], [t.join() for t in threads]
    assert not errors


def test_stream_override_and_deferred_blocks_restore_state():
    """Host-side plumbing of the LITE stream overlap: `_lib.use_stream` hands the native entry points another stream only
    inside its block (nested blocks restore the outer one), and the extractor's `deferred_stats` / `persistent_buffers`
    blocks set and clear their markers, refuse to nest, and track one busy tape per key."""
    from orbit_dataset_amd import _lib
    from orbit_dataset_amd.model.feature_extractors import create_feature_extractor

    class FakeStream:
        def __init__(self, h):
            self.cuda_stream = h

    assert getattr(_lib._tls, "stream", None) is None
    with _lib.use_stream(FakeStream(0x1234)):
        assert _lib.stream_handle().value == 0x1234
        with _lib.use_stream(FakeStream(0x5678)):
            assert _lib.stream_handle().value == 0x5678
        assert _lib.stream_handle().value == 0x1234
        # the override is per THREAD (as torch's current stream is): the prefetcher's staging thread asks for ITS stream while
        # the main thread sits inside a LITE use_stream block (a process-global override sent the staging thread's
        # uint8 conversion kernel to the LITE side stream, unordered against the upload it reads)
        import threading
        seen = {}

        def other_thread():
            seen["inside"] = getattr(_lib._tls, "stream", None)
            with _lib.use_stream(FakeStream(0x9abc)):
                seen["own"] = _lib.stream_handle().value
            seen["after"] = getattr(_lib._tls, "stream", None)
        th = threading.Thread(target=other_thread)
        th.start(), th.join()
        assert seen == {"inside": None, "own": 0x9abc, "after": None}
        assert _lib.stream_handle().value == 0x1234  # ... and the other thread's block did not touch this thread's override
    assert getattr(_lib._tls, "stream", None) is None

    fe, _ = create_feature_extractor("resnet18", False, False, True)
    assert fe._defer_stats is None and fe._persist_key is None
    with fe.deferred_stats(fe) as d:
        assert fe._defer_stats is d.pending == []
        with pytest.raises(RuntimeError):
            with fe.deferred_stats(fe):
                pass
        fe._defer_stats = d.pending  # (the failed nesting attempt's __exit__ cleared the marker)
    assert fe._defer_stats is None
    with fe.persistent_buffers(fe, "k"):
        assert fe._persist_key == "k" and fe.persistent_available("k")
        fe._persist_busy["k"] = True
        assert not fe.persistent_available("k")
    assert fe._persist_key is None
    fe.persistent_release("k")
    assert fe.persistent_available("k")
    t = fe._persistent_tensor("k", "tape", 1000, torch.uint8, torch.device("cpu"))
    assert t.numel() == 1000 and fe._persistent_tensor("k", "tape", 600, torch.uint8, torch.device("cpu")).data_ptr() == t.data_ptr()
