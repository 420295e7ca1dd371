"""End-to-end GPU parity of SingleStepFewShotRecogniser.personalise()/predict() against the CPU oracle flows
(oracle/recogniser.py, themselves pinned to the imported reference by the golden fixtures G5/G6).

The bar is BASELINE.json's: logits within 1e-3 (absolute, fp32) and identical per-frame argmax, hence identical
frame accuracy (reference utils/eval_metrics.py:27-36)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from oracle import blocks  # noqa: E402
from oracle.recogniser import OracleRecogniser  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.data.utils import attach_frame_history  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402

LOGIT_TOL = 1e-3


def build_pair(fe_name, adapt, classifier, clip_length, batch_size, num_lite=16, logit_scale=1.0):
    model = SingleStepFewShotRecogniser(fe_name, adapt, classifier, clip_length, batch_size, False, num_lite,
                                        logit_scale)
    # FiLM modulation depth: ~+-10% for resnet18; the random efficientnet_b0 is far more sensitive (16 gated blocks)
    synthetic.init_parameters_(model, film_strength=0.02 if fe_name == "efficientnet_b0" else 0.1)
    if adapt:  # gamma0/beta0 are snapshotted at construction in the reference; refresh after loading parameters
        from orbit_dataset_amd.model.film import get_film_parameters
        model.film_generator.initial_film_parameters = get_film_parameters(model.film_parameter_names,
                                                                           model.feature_extractor)
    model._set_device("cuda:0")
    model._send_to_device()
    model.set_test_mode(True)
    ref = OracleRecogniser(fe_name, adapt, classifier, clip_length, batch_size, num_lite, logit_scale)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v for k, v in sd.items() if k.startswith("feature_extractor.")})
    if adapt:
        ref.set_encoder.load_state_dict({k[len("set_encoder."):]: v for k, v in sd.items() if k.startswith("set_encoder.")})
        gen = ref.build_film_generator()
        gen.load_state_dict({k[len("film_generator."):]: v for k, v in sd.items() if k.startswith("film_generator.")})
    return model, ref


def check_task(model, ref, task, to_device):
    ctx, lab, tgt = task["context_clips"], task["context_labels"], task["target_clips"]
    with torch.no_grad():  # every test-time caller of the reference does (single-step-learner.py:249,311; notebook)
        if to_device:
            model.personalise(ctx.cuda(), lab.cuda())
            logits = model.predict(tgt.cuda())
        else:  # frames stay on the host and are moved per mini-batch, as in the reference's test loop
            model.personalise(ctx, lab.cuda())
            logits = model.predict(tgt)
    ref.personalise(ctx, lab)
    want = ref.predict(tgt)
    got = logits.cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err < LOGIT_TOL, f"max |dlogit| = {err}"
    assert torch.equal(got.argmax(1), want.argmax(1))
    assert blocks.frame_accuracy(got, task["target_labels"]) == blocks.frame_accuracy(want, task["target_labels"])
    model._reset()
    ref.reset()
    return err


@pytest.mark.parametrize("classifier,scale", [("proto", 1.0), ("proto_cosine", 32.0)])
def test_config2_resnet18_84(device, classifier, scale):
    """BASELINE config 2 shape at reduced count (oracle finishes in seconds): 5-way, 84x84, batch chunking."""
    model, ref = build_pair("resnet18", False, classifier, 1, 32, logit_scale=scale)
    task = synthetic.make_task(0, way=5, shots=2, frames_per_shot=8, num_query=50, frame_size=84)
    check_task(model, ref, task, to_device=True)
    task = synthetic.make_task(1, way=5, shots=1, frames_per_shot=1, num_query=20, frame_size=84)  # config 1: 1-shot
    check_task(model, ref, task, to_device=False)


def test_clip_pooling_and_frame_history(device):
    """clip_length 8 (pooler) on support; query clips built by attach_frame_history as in the test loop
    (single-step-learner.py:327-334)."""
    model, ref = build_pair("resnet18", False, "proto", 4, 3)
    task = synthetic.make_task(2, way=3, shots=2, frames_per_shot=4, num_query=1, frame_size=64, clip_length=4,
                               label_values=(3, 7, 9))
    frames = synthetic.make_task(3, way=3, shots=1, frames_per_shot=1, num_query=10, frame_size=64)["target_clips"][:, 0]
    clips = attach_frame_history(frames, 4)
    assert torch.equal(clips, blocks.attach_frame_history(frames, 4))
    task["target_clips"] = clips
    task["target_labels"] = torch.full((10,), 7)
    check_task(model, ref, task, to_device=True)


@pytest.mark.parametrize("fe_name,size", [("resnet18", 84), ("efficientnet_b0", 224)])
def test_config4_cnaps_adaptation(device, fe_name, size):
    """adapt_features=True: set encoder -> task embedding -> FiLM generator -> FiLM-modulated extractor."""
    model, ref = build_pair(fe_name, True, "proto", 1, 16)
    task = synthetic.make_task(4, way=5, shots=1, frames_per_shot=4, num_query=16, frame_size=size)
    check_task(model, ref, task, to_device=True)
    assert float(model.film_generator.regularization_term()) > 0


@pytest.mark.parametrize("classifier", ["versa", "mahalanobis"])
def test_cnaps_and_simple_cnaps_heads_end_to_end(device, classifier):
    """The README's CNAPs (versa) and Simple CNAPs (mahalanobis) recipes: FiLM-adapted extractor + the head, against the
    oracle's restatement of the head applied to the same (natively extracted) features."""
    model = SingleStepFewShotRecogniser("resnet18", True, classifier, 1, 16, False, 16, 1.0)
    synthetic.init_parameters_(model)
    from orbit_dataset_amd.model.film import get_film_parameters
    model.film_generator.initial_film_parameters = get_film_parameters(model.film_parameter_names, model.feature_extractor)
    model._set_device("cuda:0")
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(9, way=4, shots=2, frames_per_shot=5, num_query=24, frame_size=84)
    ctx, lab, tgt = task["context_clips"].cuda(), task["context_labels"].cuda(), task["target_clips"].cuda()
    with torch.no_grad():
        model.personalise(ctx, lab)
        logits = model.predict(tgt).cpu()
        fc = model._get_features_in_batches(ctx, model.film_dict).cpu().double()
        fq = model._get_features_in_batches(tgt, model.film_dict).cpu().double()
    labels = task["context_labels"]
    if classifier == "versa":
        wp, bp = blocks.DenseResidualBlock(512, 512), blocks.DenseResidualBlock(512, 1)
        wp.load_state_dict({k: v.cpu() for k, v in model.classifier.weight_processor.state_dict().items()})
        bp.load_state_dict({k: v.cpu() for k, v in model.classifier.bias_processor.state_dict().items()})
        with torch.no_grad():
            _, W, b = blocks.versa_configure(fc, labels, wp.double(), bp.double())
        want = fq @ W.t() + b
    else:
        _, means, precisions, _, _ = blocks.mahalanobis_configure(fc, labels)
        want = blocks.mahalanobis_predict(fq, means, precisions)
    assert (logits.double() - want).abs().max().item() < 1e-3 * want.abs().max().item()
    assert torch.equal(logits.argmax(1), want.argmax(1))
    model._reset()
    with pytest.raises(AttributeError):
        model.predict(tgt)


@pytest.mark.parametrize("fe_name,adapt,size", [("resnet18", False, 84), ("resnet18", True, 84),
                                                ("efficientnet_b0", False, 128)])
def test_overlap_query_stream_is_bit_identical(device, fe_name, adapt, size):
    """overlap_query: the query pass of predict() on a second stream, concurrent with the support pass. Same kernels, same
    inputs -> bit-identical logits, for resident and for host-side clips, over repeated tasks."""
    model, _ = build_pair(fe_name, adapt, "proto", 1, 16)
    tasks = [synthetic.make_task(20 + i, way=4, shots=2, frames_per_shot=6, num_query=40, frame_size=size) for i in range(3)]

    def run(overlap, on_device):
        model.overlap_query = overlap
        outs = []
        with torch.no_grad():
            for t in tasks:
                ctx, tgt = t["context_clips"], t["target_clips"]
                if on_device:
                    ctx, tgt = ctx.cuda(), tgt.cuda()
                    torch.cuda.synchronize()  # the contract of overlap_query: resident query clips are ready
                model.personalise(ctx, t["context_labels"].cuda())
                outs.append(model.predict(tgt))
                model._reset()
        torch.cuda.synchronize()
        return [o.cpu() for o in outs]

    base = run(False, True)
    for on_device in (True, False):
        got = run(True, on_device)
        assert all(torch.equal(a, b) for a, b in zip(base, got))
    model.overlap_query = False


def test_default_overlap_mode_is_safe_and_bit_identical(device):
    """Round 6: the recogniser's DEFAULT is overlap_query = "auto" - the query pass of predict() runs on the second stream
    exactly when the clips are known to be ready: host-resident clips, or device tensors marked with data.utils.mark_ready
    (TaskPrefetcher marks what it yields). An unmarked device tensor - possibly the product of work still pending on the
    caller's stream - takes the serial order. Logits are bit-identical in all three cases, also when the query clips are
    PRODUCED on the caller's stream after personalise() was queued (the case the opt-in forms must not be used for)."""
    from orbit_dataset_amd.data.utils import mark_ready, ready_event
    model, _ = build_pair("resnet18", False, "proto", 1, 16)
    assert model.overlap_query == "auto"
    t = synthetic.make_task(31, way=4, shots=2, frames_per_shot=6, num_query=40, frame_size=84)
    ctx, lab, tgt = t["context_clips"].cuda(), t["context_labels"].cuda(), t["target_clips"].cuda()
    torch.cuda.synchronize()

    def run(q):
        with torch.no_grad():
            model.personalise(ctx, lab)
            out = model.predict(q() if callable(q) else q).clone()
        model._reset()
        return out
    model.overlap_query = False
    base = run(tgt)
    model.overlap_query = "auto"
    used = lambda: model.__dict__.get("_query_stream") is not None
    assert not used()
    # unmarked device clips produced on the caller's stream AFTER personalise() was queued: serial order, correct values
    late = lambda: (tgt * 2.0) * 0.5  # (exact in fp32: the same values, but a tensor that does not exist before this point)
    assert torch.equal(run(late), base) and not used()
    # marked clips: second stream, from the readiness event
    marked = mark_ready(tgt.clone())
    assert ready_event(marked) is not None and ready_event(marked[:10]) is None  # the mark lives on the tensor object
    assert torch.equal(run(marked), base) and used()
    # a mark recorded right after the producing kernel, consumed while that kernel's stream is still busy
    def produced_then_marked():
        return mark_ready((tgt * 2.0) * 0.5)
    assert torch.equal(run(produced_then_marked), base)
    # host-resident clips: their upload is issued on the second stream
    assert torch.equal(run(t["target_clips"]), base)
    torch.cuda.synchronize()


def test_predict_video_equals_predict_on_frame_history(device):
    """predict_video: each frame through the extractor once, windows pooled on the features — bit-identical to the
    reference's attach_frame_history -> predict on the T-times larger clip tensor."""
    from orbit_dataset_amd.data.utils import attach_frame_history
    for T in (3, 8):
        model, _ = build_pair("resnet18", False, "proto", T, 4)
        t = synthetic.make_task(30 + T, way=3, shots=1, frames_per_shot=2 * T, num_query=1, frame_size=64, clip_length=T)
        video = synthetic.make_task(31, way=3, shots=1, frames_per_shot=1, num_query=11, frame_size=64)["target_clips"][:, 0]
        with torch.no_grad():
            model.personalise(t["context_clips"].cuda(), t["context_labels"].cuda())
            want = model.predict(attach_frame_history(video.cuda(), T))
            got = model.predict_video(video.cuda())
        assert got.shape == want.shape == (11, 3) and torch.equal(got, want)


def test_uint8_clips_match_host_side_normalisation(device):
    """8-bit clips are normalised on the GPU exactly as the reference's loader does on the host: same logits."""
    from orbit_dataset_amd.data.utils import NORMALIZE_STATS
    model, _ = build_pair("resnet18", False, "proto", 1, 16)
    g = torch.Generator().manual_seed(5)
    ctx8 = torch.randint(0, 256, (12, 1, 3, 64, 64), generator=g, dtype=torch.uint8)
    tgt8 = torch.randint(0, 256, (9, 1, 3, 64, 64), generator=g, dtype=torch.uint8)
    labels = torch.arange(3).repeat_interleave(4)
    mean, std = (torch.tensor(v)[None, None, :, None, None] for v in NORMALIZE_STATS["imagenet"])
    norm = lambda u8: (u8.float().div(255) - mean) / std
    with torch.no_grad():
        model.personalise(norm(ctx8), labels.cuda())
        want = model.predict(norm(tgt8))
        model._reset()
        model.personalise(ctx8.pin_memory(), labels.cuda())
        got = model.predict(tgt8)
    assert torch.equal(got, want)


def test_config3_efficientnet_224(device):
    model, ref = build_pair("efficientnet_b0", False, "proto", 1, 16)
    task = synthetic.make_task(5, way=5, shots=1, frames_per_shot=4, num_query=12, frame_size=224)
    check_task(model, ref, task, to_device=True)


def test_lite_forward(device):
    """personalise_with_lite / predict_a_batch forward semantics: permutation from np.random, caches, label
    reordering (few_shot_recognisers.py:328-343,388-437; learner loop single-step-learner.py:212-243)."""
    model, ref = build_pair("resnet18", True, "proto", 1, 8, num_lite=4)
    task = synthetic.make_task(6, way=4, shots=1, frames_per_shot=5, num_query=12, frame_size=64)
    ctx, lab, tgt = task["context_clips"], task["context_labels"], task["target_clips"]
    model._clear_caches()
    ref.clear_caches()
    for b in range(2):
        np.random.seed(100 + b)
        model.personalise_with_lite(ctx.cuda(), lab.cuda())
        got = model.predict_a_batch(tgt[b * 6:(b + 1) * 6].cuda()).cpu()
        np.random.seed(100 + b)
        ref.personalise_with_lite(ctx, lab)
        ref_clips = tgt[b * 6:(b + 1) * 6]
        want = blocks.proto_predict(blocks.mean_pool(ref._features(ref_clips, ref.film_dict), 1), ref.W, ref.b)
        assert (got - want).abs().max().item() < LOGIT_TOL
        assert torch.equal(got.argmax(1), want.argmax(1))
        model._reset()
        ref.reset()


def test_errors_match_reference(device):
    with pytest.raises(ValueError):
        SingleStepFewShotRecogniser("resnet18", False, "nope", 1, 8, False, 4)
    with pytest.raises(ValueError):
        SingleStepFewShotRecogniser("vgg", False, "proto", 1, 8, False, 4)
    model, _ = build_pair("resnet18", False, "proto", 1, 8)
    with pytest.raises(AttributeError):  # predict before personalise (classifier_heads.py:210-211)
        model.predict(torch.zeros(2, 1, 3, 32, 32, device=device))


def test_learner_test_mode_end_to_end(device, tmp_path):
    """single-step-learner counterpart, --mode test: personalise -> per video {attach_frame_history -> predict} ->
    _reset, with clip_length 2, on three synthetic tasks."""
    from orbit_dataset_amd.learner import main
    out = tmp_path / "results.json"
    stats = main(["--mode", "test", "--feature_extractor", "resnet18", "--classifier", "proto", "--frame_size", "64",
                  "--clip_length", "2", "--batch_size", "16", "--way", "3", "--shots", "2", "--frames_per_shot", "4",
                  "--num_query_videos", "2", "--frames_per_video", "6", "--num_test_tasks", "3",
                  "--results_path", str(out)])["test"]
    assert stats["num_tasks"] == 3 and 0.0 <= stats["frame_acc"][0] <= 1.0
    assert stats["personalise_ms"][0] > 0 and stats["inference_ms_per_frame"][0] > 0
    assert out.exists()
    # --mode train: LITE meta-training on the native backward kernels
    common = ["--frame_size", "64", "--way", "3", "--shots", "2", "--frames_per_shot", "4", "--num_query_videos", "2",
              "--frames_per_video", "6", "--batch_size", "8", "--num_lite_samples", "4", "--tasks_per_batch", "2",
              "--num_train_tasks", "4"]
    for extra in (["--learn_extractor", "--with_lite"], ["--adapt_features", "--with_lite"], ["--learn_extractor"],
                  ["--adapt_features", "--with_lite", "--classifier", "versa"],          # README: CNAPs
                  ["--adapt_features", "--with_lite", "--classifier", "mahalanobis"]):   # README: Simple CNAPs
        tr = main(["--mode", "train", "--feature_extractor", "resnet18"] + extra + common)["train"]
        assert tr["num_tasks"] == 4 and np.isfinite(tr["loss"][0]) and tr["loss"][0] > 0
    tr = main(["--mode", "train", "--learn_extractor", "--with_lite", "--feature_extractor", "efficientnet_b0"] + common)
    assert tr["train"]["num_tasks"] == 4 and np.isfinite(tr["train"]["loss"][0])
    # validate() after every epoch (reference single-step-learner.py:245-296): per-video frame accuracy on held-out tasks, the
    # best model so far is saved - strictly-greater rule, so the history's `better` flags follow the running maximum
    best = tmp_path / "best.pt"
    tr = main(["--mode", "train", "--feature_extractor", "resnet18", "--learn_extractor", "--with_lite", "--epochs", "3",
               "--num_val_tasks", "2", "--learning_rate", "1e-3", "--save_best_model_path", str(best)] + common)["train"]
    hist = tr["validation"]
    assert [h[0] for h in hist] == [1, 2, 3] and all(0.0 <= h[1] <= 1.0 for h in hist)
    running = 0.0
    for _, acc, better in hist:
        assert better == (acc > running)
        running = max(running, acc)
    assert tr["best_validation"][0] == running and (best.exists() == (running > 0.0))
    if best.exists():
        sd = torch.load(str(best))
        assert any(k.startswith("feature_extractor.") for k in sd)
    # train_test tests the final model AND the best-validation checkpoint (reference run(), single-step-learner.py:186-188)
    best2 = tmp_path / "best2.pt"
    both = main(["--mode", "train_test", "--feature_extractor", "resnet18", "--learn_extractor", "--with_lite", "--epochs", "2",
                 "--num_val_tasks", "2", "--num_test_tasks", "2", "--learning_rate", "1e-3",
                 "--save_best_model_path", str(best2)] + common)
    assert "test" in both and both["test"]["num_tasks"] == 2
    if both["train"]["best_validation"] is not None and best2.exists():
        assert both["test_best_validation"]["num_tasks"] == 2 and 0.0 <= both["test_best_validation"]["frame_acc"][0] <= 1.0
    else:
        assert "test_best_validation" not in both


def test_sharded_forms_on_device_world1(device):
    """dist.personalise_support_sharded / predict_query_sharded executed on the GPU (world 1: the all-reduce is the
    identity) must reproduce personalise()/predict() exactly, including with FiLM adaptation — the partial-sum payload,
    global label set and embedding-sum exchange are the code the N>1 run uses."""
    from orbit_dataset_amd import dist as odist
    for adapt in (False, True):
        model, _ = build_pair("resnet18", adapt, "proto", 1, 8)
        task = synthetic.make_task(9, way=4, shots=1, frames_per_shot=5, num_query=11, frame_size=64,
                                   label_values=(2, 5, 6, 9))
        ctx, lab, tgt = task["context_clips"].cuda(), task["context_labels"].cuda(), task["target_clips"].cuda()
        model.personalise(ctx, lab)
        want_W, want = model.classifier.weight.clone(), model.predict(tgt)
        model._reset()
        sh = odist.SupportSharding(0, 1)
        odist.personalise_support_sharded(model, ctx, lab, sh)
        assert torch.allclose(model.classifier.weight, want_W, atol=1e-6)
        got = odist.predict_query_sharded(model, tgt, sh)
        assert torch.allclose(got, want, atol=1e-5)
        # emulate rank 1 of 2 locally: its slice alone must give partial sums that add up with rank 0's
        payloads = []
        for r in range(2):
            s2 = odist.SupportSharding(r, 2)
            s2.reduce_ = lambda t: payloads.append(t.clone()) or t  # capture instead of all-reduce
            odist.personalise_support_sharded(model, ctx, lab, s2)
            model._reset()
        if not adapt:
            C, D = 4, 512
            total = payloads[0] + payloads[1]
            W = 2 * total[:C * D].reshape(C, D) / total[C * D:, None]
            assert torch.allclose(W, want_W, atol=1e-5)


def test_edge_empty_query_and_ragged_batches(device):
    """empty query set -> [0, C] logits; support/query counts that leave a ragged last mini-batch; single clip class."""
    model, ref = build_pair("resnet18", False, "proto", 1, 7)  # batch_size 7: 20 support clips -> 7 + 7 + 6
    task = synthetic.make_task(12, way=5, shots=1, frames_per_shot=4, num_query=15, frame_size=64)
    check_task(model, ref, task, to_device=True)
    model.personalise(task["context_clips"].cuda(), task["context_labels"].cuda())
    empty = model.predict(task["target_clips"][:0].cuda())
    assert tuple(empty.shape) == (0, 5)
    model._reset()
    one = synthetic.make_task(13, way=1, shots=1, frames_per_shot=3, num_query=4, frame_size=64)
    model.personalise(one["context_clips"].cuda(), one["context_labels"].cuda())
    assert tuple(model.predict(one["target_clips"].cuda()).shape) == (4, 1)


@pytest.mark.parametrize("adapt", [False, True])
def test_pipelined_overlap_is_bit_identical_over_consecutive_tasks(device, adapt):
    """overlap_query = 2: extractor + head of predict() on the second stream, not joined, so the next task's personalise()
    starts while this task's query pass runs. Same kernels on the same inputs: bit-identical logits for every task of a
    back-to-back sequence. What the second stream still reads must survive _reset() / the next personalise() on the caller's
    stream: the class weights and, with adapt_features (CNAPs / FiLM), this task's generated gamma / beta vectors - between
    the tasks the caller's stream allocates and overwrites scratch of the same sizes, so a block handed back too early shows
    up as different logits."""
    model = SingleStepFewShotRecogniser("resnet18", adapt, "proto", 1, 16, False, 16, 1.0)
    synthetic.init_parameters_(model)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(True)
    tasks = [synthetic.make_task_on_device(50 + i, 5, 1, 6, 40, 64, 1, device) for i in range(6)]

    def run(mode):
        model.overlap_query = mode
        outs = []
        with torch.no_grad():
            for _ in range(2):
                for t in tasks:
                    model.personalise(t["context_clips"], t["context_labels"])
                    outs.append(model.predict(t["target_clips"]))
                    model._reset()
                    if mode == 2:  # churn the caller stream's allocator with blocks of the freed sizes
                        junk = [torch.full((n,), float("nan"), device=device) for n in (64, 128, 256, 512, 5 * 512, 4800)]
                        del junk
        torch.cuda.synchronize()
        model.overlap_query = False
        return [o.clone() for o in outs]

    base, piped = run(False), run(2)
    assert len(piped) == 12 and model.logits_ready is not None
    for a, b in zip(base, piped):
        assert torch.equal(a, b)
