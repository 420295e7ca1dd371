"""Generate the golden fixtures that pin oracle/ to the REFERENCE's own code.

Run once in the build container (the reference is mounted read-only there and never travels):

    python tests/golden/make_golden.py [/root/reference]

The reference's modules are imported as they are (`model.classifier_heads`, `model.poolers`,
`model.set_encoders`, `model.feature_adapters`, `data.utils`), and `model.few_shot_recognisers` is imported
with the absent third-party `timm` package stubbed in sys.modules and `create_feature_extractor` replaced by a
factory returning this build's PyTorch-CPU extractor (the extractor's layer arithmetic is pinned separately,
against Hugging Face transformers: make_golden_hf.py / G12). Inputs come from the deterministic synthetic generators; inputs and the reference's
outputs are written to small .npz files next to this script. Only data is stored — no reference source.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import orbit_dataset_amd  # noqa: E402,F401
from oracle import extractors as oracle_extractors  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402


def stub_timm():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Absent:  # isinstance() targets of model/film.py; never instantiated
        pass

    def _ctor(*a, **k):
        raise RuntimeError("timm is not available offline")

    mod("timm")
    mod("timm.models")
    mod("timm.models.registry", get_pretrained_cfg=lambda name: {})
    mod("timm.models.efficientnet", EfficientNet=_Absent, tf_efficientnet_b0=_ctor, tf_efficientnetv2_s_in21k=_ctor)
    mod("timm.models.efficientnet_blocks", ConvBnAct=_Absent, InvertedResidual=_Absent, CondConvResidual=_Absent,
        EdgeResidual=_Absent)
    mod("timm.models.vision_transformer", vit_small_patch32_224_in21k=_ctor, vit_base_patch32_224_in21k=_ctor,
        vit_base_patch32_224_clip_laion2b=_ctor)
    mod("timm.scheduler", create_scheduler=_ctor)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %6.1f KB  %s" % (name, os.path.getsize(path) / 1024, sorted(out)))


def g1_head():
    from model.classifier_heads import PrototypicalClassifier
    cases = {}
    g = torch.Generator().manual_seed(101)
    specs = [("w5_d512", 50, 40, 512, list(range(5))), ("w10_d96", 50, 40, 96, list(range(10))),
             ("noncontig_d96", 30, 20, 96, [3, 7, 9]), ("oneshot_d64", 5, 12, 64, list(range(5))),
             # the extractor width of the headline (efficientnet_b0: D = 1280), 5- and 10-way (BASELINE configs 3 and 5);
             # appended so that the generator state of the four cases above, and with it their recorded values, is unchanged
             ("w5_d1280", 50, 40, 1280, list(range(5))), ("w10_d1280", 50, 40, 1280, list(range(10)))]
    for name, N, M, D, values in specs:
        way = len(values)
        cls = torch.arange(way).repeat_interleave(N // way)[torch.randperm(N, generator=g)]
        centers = torch.randn(way, D, generator=g) * 0.3
        feats = torch.relu(centers[cls] + 0.5 * torch.randn(N, D, generator=g))
        labels = torch.tensor(values)[cls]
        q = torch.relu(centers[torch.randint(0, way, (M,), generator=g)] + 0.5 * torch.randn(M, D, generator=g))
        q[0] = 0  # zero query row: cosine logits must come out as exactly 0
        cases[name + "_feats"], cases[name + "_labels"], cases[name + "_q"] = feats, labels, q
        for dist in ("euclidean", "cosine"):
            for scale in (1.0, 32.0):
                head = PrototypicalClassifier(scale, dist)
                head.configure(feats, labels)
                logits = head.predict(q)
                key = "%s_%s_s%d" % (name, dist, int(scale))
                cases[key + "_W"] = head.weight
                if dist == "euclidean":
                    cases[key + "_b"] = head.bias
                cases[key + "_logits"] = logits
        # reference quirk: configured weights are re-wrapped in nn.Parameter -> no grad to support
        f2 = feats.clone().requires_grad_(True)
        head = PrototypicalClassifier(1.0, "euclidean")
        head.configure(f2, labels)
        head.predict(q).sum().backward()
        cases[name + "_support_grad_is_none"] = np.array(f2.grad is None)
    save("G1_head", **cases)


def g2_pooler():
    from model.poolers import MeanPooler
    g = torch.Generator().manual_seed(102)
    x = torch.randn(24, 40, generator=g)
    save("G2_pooler", x=x, T1=MeanPooler(1)(x), T8=MeanPooler(8)(x), T3=MeanPooler(3)(x))


def g3_set_encoder():
    from model.set_encoders import SetEncoder
    enc = SetEncoder()
    synthetic.init_parameters_(enc)
    x = synthetic.make_task(7, way=2, shots=1, frames_per_shot=2, num_query=1, frame_size=84)["context_clips"]
    enc.eval()
    with torch.no_grad():
        reps = enc(x)
        mean = enc.aggregate([reps[:2], reps[2:]], "mean")
        x32 = x[:, :, :, :32, :32]
        reps32 = enc(x32)
    save("G3_set_encoder", x=x, reps=reps, mean=mean, reps32=reps32)


def g4_film_generator():
    from model.feature_adapters import FilmParameterGenerator
    fe = oracle_extractors.create("efficientnet_b0")
    synthetic.init_parameters_(fe)
    names = []
    for n in fe.film_slot_names():
        names += [n + ".weight", n + ".bias"]
    params = dict(fe.named_parameters())
    sizes = {n: len(params[n]) for n in names}
    initial = {n: params[n].detach().clone() for n in names}
    gen = FilmParameterGenerator(sizes, initial, pooled_size=64, hidden_size=64)
    synthetic.init_parameters_(gen, prefix="film_generator.")
    z = torch.randn(1, 64, generator=torch.Generator().manual_seed(104))
    with torch.no_grad():
        film = gen(z)
    out = {"z": z, "l2_term": gen.regularization_term(), "names": np.array(gen.film_parameter_names)}
    for i, n in enumerate(gen.film_parameter_names):
        out["film_%03d" % i] = film[n]
    save("G4_film_generator", **out)


def make_reference_recogniser(fsr, fe_name, adapt, classifier, clip_length, batch_size, num_lite, logit_scale=1.0,
                              film_strength=0.1, learn_extractor=False):
    def factory(feature_extractor_name, pretrained, with_film=False, learn_extractor=True):
        fe = oracle_extractors.create(feature_extractor_name)
        synthetic.init_parameters_(fe)
        if not learn_extractor:
            for p in fe.parameters():
                p.requires_grad = False
        names = None
        if with_film:
            mods = dict(fe.named_modules())
            for n in fe.film_slot_names():
                mods[n].film = True  # what the reference's tag_film_layers does (model/film.py:54-55)
            from model.film import get_film_parameter_names
            names = get_film_parameter_names(feature_extractor_name, fe)
        return fe, names

    fsr.create_feature_extractor = factory
    model = fsr.SingleStepFewShotRecogniser(fe_name, adapt, classifier, clip_length, batch_size, learn_extractor,
                                            num_lite, logit_scale)
    sd = synthetic.synthetic_state_dict(model, film_strength=film_strength)
    model.load_state_dict(sd)
    if adapt:  # the reference snapshots gamma0/beta0 at construction (film.py:81-87): they already hold the
        pass   # synthetic values because the factory initialises the extractor before FilmParameterGenerator is built
    model._set_device(torch.device("cpu"))
    model._send_to_device()
    return model


def g5_recogniser(fsr):
    from data.utils import attach_frame_history
    out = {}
    task = synthetic.make_task(21, way=5, shots=1, frames_per_shot=2, num_query=12, frame_size=32)
    out["context_clips"], out["context_labels"] = task["context_clips"], task["context_labels"]
    out["target_clips"] = task["target_clips"]
    for tag, adapt, classifier, scale in (("proto", False, "proto", 1.0), ("cosine", False, "proto_cosine", 32.0),
                                          ("film", True, "proto", 1.0)):
        model = make_reference_recogniser(fsr, "resnet18", adapt, classifier, 1, 4, 16, scale)
        model.set_test_mode(True)
        with torch.no_grad():
            model.personalise(task["context_clips"], task["context_labels"])
            out[tag + "_logits"] = model.predict(task["target_clips"])
            out[tag + "_W"] = model.classifier.weight
            if adapt:
                out[tag + "_l2"] = model.film_generator.regularization_term()
                out[tag + "_film_bn1_weight"] = model.film_dict["bn1.weight"]
        model._reset()
    # clip_length 3: support clips of 3 frames, query video expanded with attach_frame_history (learner :327-334)
    t3 = synthetic.make_task(22, way=3, shots=2, frames_per_shot=3, num_query=1, frame_size=32, clip_length=3,
                             label_values=(3, 7, 9))
    video = synthetic.make_task(23, way=3, shots=1, frames_per_shot=1, num_query=7, frame_size=32)["target_clips"][:, 0]
    clips = attach_frame_history(video, 3)
    model = make_reference_recogniser(fsr, "resnet18", False, "proto", 3, 2, 16)
    model.set_test_mode(True)
    with torch.no_grad():
        model.personalise(t3["context_clips"], t3["context_labels"])
        out["T3_logits"] = model.predict(clips)
    out["T3_context_clips"], out["T3_context_labels"], out["T3_video"], out["T3_clips"] = (
        t3["context_clips"], t3["context_labels"], video, clips)
    save("G5_recogniser", **out)


def g6_lite(fsr):
    """LITE meta-training step exactly as Learner.train_task_with_lite drives it (single-step-learner.py:212-243):
    frozen extractor, learnable set encoder + FiLM generator (the README's adapt_features recipe)."""
    import torch.nn.functional as F
    task = synthetic.make_task(31, way=4, shots=1, frames_per_shot=3, num_query=8, frame_size=32)
    model = make_reference_recogniser(fsr, "resnet18", True, "proto", 1, 4, 3)
    model.set_test_mode(False)
    num_lite, tasks_per_batch, batch_size = 3, 2, 4
    ctx, lab, tgt, tlab = task["context_clips"], task["context_labels"], task["target_clips"], task["target_labels"]
    out = {"context_clips": ctx, "context_labels": lab, "target_clips": tgt, "target_labels": tlab,
           "num_lite_samples": num_lite, "tasks_per_batch": tasks_per_batch, "batch_size": batch_size}
    model._clear_caches()
    model.zero_grad()
    for b in range(2):
        np.random.seed(500 + b)
        out["perm_%d" % b] = np.random.permutation(len(ctx))
        np.random.seed(500 + b)
        model.personalise_with_lite(ctx, lab)
        logits = model.predict_a_batch(tgt[b * batch_size:(b + 1) * batch_size])
        scaling = len(lab) / (num_lite * tasks_per_batch)
        loss = scaling * F.cross_entropy(logits, tlab[b * batch_size:(b + 1) * batch_size])
        loss = loss + 0.001 * model.film_generator.regularization_term()
        loss.backward()
        out["logits_%d" % b], out["loss_%d" % b] = logits, loss
        out["l2_%d" % b] = model.film_generator.regularization_term()
        model._reset()
    params = dict(model.named_parameters())
    for name in ("set_encoder.encoder.layer1.0.weight", "set_encoder.encoder.layer5.1.bias",
                 "film_generator.regularizers.0", "film_generator.generators.0.block.3.bias"):
        out["grad__" + name] = params[name].grad
    out["extractor_has_grad"] = np.array(any(p.grad is not None for p in model.feature_extractor.parameters()))
    save("G6_lite", **out)


def g8_lite_learn_extractor(fsr):
    """LITE meta-training steps with an UNFROZEN extractor (the README's `--learn_extractor --with_lite` recipe):
    the extractor runs BatchNorm in train() mode (few_shot_recognisers.py:176-183) on every pass, including the
    no-grad cache passes, and loss.backward() fills the extractor's .grad. Variant a: ProtoNets; variant b: + FiLM."""
    import torch.nn.functional as F
    task = synthetic.make_task(41, way=4, shots=1, frames_per_shot=4, num_query=16, frame_size=64)
    ctx, lab, tgt, tlab = task["context_clips"], task["context_labels"], task["target_clips"], task["target_labels"]
    num_lite, tasks_per_batch, batch_size = 4, 2, 8
    out = {"context_clips": ctx, "context_labels": lab, "target_clips": tgt, "target_labels": tlab,
           "num_lite_samples": num_lite, "tasks_per_batch": tasks_per_batch, "batch_size": batch_size}
    watched = ("feature_extractor.conv1.weight", "feature_extractor.bn1.weight", "feature_extractor.bn1.bias",
               "feature_extractor.layer1.0.conv1.weight", "feature_extractor.layer1.1.bn2.weight",
               "feature_extractor.layer2.0.conv1.weight", "feature_extractor.layer2.0.downsample.0.weight",
               "feature_extractor.layer2.0.downsample.1.bias", "feature_extractor.layer3.1.conv2.weight",
               "feature_extractor.layer4.0.conv2.weight", "feature_extractor.layer4.1.bn2.bias",
               "set_encoder.encoder.layer1.0.weight", "film_generator.regularizers.0",
               "film_generator.generators.3.block.0.weight")
    stats = ("feature_extractor.bn1.running_mean", "feature_extractor.bn1.running_var",
             "feature_extractor.layer2.0.downsample.1.running_mean", "feature_extractor.layer4.1.bn2.running_var",
             "feature_extractor.layer4.1.bn2.num_batches_tracked")
    for tag, adapt in (("a", False), ("b", True)):
        model = make_reference_recogniser(fsr, "resnet18", adapt, "proto", 1, batch_size, num_lite,
                                          learn_extractor=True)
        model.set_test_mode(False)
        model._clear_caches()
        model.zero_grad()
        for b in range(2):
            np.random.seed(800 + b)
            model.personalise_with_lite(ctx, lab)
            logits = model.predict_a_batch(tgt[b * batch_size:(b + 1) * batch_size])
            scaling = len(lab) / (num_lite * tasks_per_batch)
            loss = scaling * F.cross_entropy(logits, tlab[b * batch_size:(b + 1) * batch_size])
            loss = loss + 0.001 * model.film_generator.regularization_term()
            loss.backward()
            out["%s_logits_%d" % (tag, b)], out["%s_loss_%d" % (tag, b)] = logits, loss
            model._reset()
        params = dict(model.named_parameters())
        for name in watched:
            if name in params and params[name].grad is not None:
                # large filters are stored as a strided sample (<= 4096 values) plus the L2 norm of the whole gradient
                flat = params[name].grad.flatten()
                out["%s_grad__%s" % (tag, name)] = flat[::max(1, flat.numel() // 4096)][:4096].clone()
                out["%s_gnorm__%s" % (tag, name)] = flat.double().norm().float()
        sd = model.state_dict()
        for name in stats:
            out["%s_stat__%s" % (tag, name)] = sd[name].float()
        # film-replaced BatchNorm weights receive no gradient under functional_call
        out["%s_bn1_weight_has_grad" % tag] = np.array(params["feature_extractor.bn1.weight"].grad is not None)
    save("G8_lite_learn_extractor", **out)


def g9_lite_efficientnet(fsr):
    """The README's main recipe (`--feature_extractor efficientnet_b0 --learn_extractor --with_lite`): one LITE step of
    the reference with the tf_efficientnet_b0-layout extractor unfrozen (train-mode BatchNorm), variant b adds FiLM."""
    import torch.nn.functional as F
    task = synthetic.make_task(51, way=3, shots=1, frames_per_shot=3, num_query=6, frame_size=64)
    ctx, lab, tgt, tlab = task["context_clips"], task["context_labels"], task["target_clips"], task["target_labels"]
    num_lite, tasks_per_batch, batch_size = 3, 2, 6
    out = {"context_clips": ctx, "context_labels": lab, "target_clips": tgt, "target_labels": tlab,
           "num_lite_samples": num_lite, "tasks_per_batch": tasks_per_batch, "batch_size": batch_size}
    watched = ("feature_extractor.conv_stem.weight", "feature_extractor.bn1.weight",
               "feature_extractor.blocks.0.0.conv_dw.weight", "feature_extractor.blocks.0.0.se.conv_reduce.weight",
               "feature_extractor.blocks.1.0.conv_pw.weight", "feature_extractor.blocks.1.1.conv_dw.weight",
               "feature_extractor.blocks.2.0.se.conv_expand.bias", "feature_extractor.blocks.3.2.bn2.weight",
               "feature_extractor.blocks.4.1.conv_pwl.weight", "feature_extractor.blocks.5.3.conv_dw.weight",
               "feature_extractor.blocks.6.0.bn3.weight", "feature_extractor.conv_head.weight",
               "feature_extractor.bn2.bias", "set_encoder.encoder.layer1.0.weight", "film_generator.regularizers.5")
    stats = ("feature_extractor.bn1.running_mean", "feature_extractor.blocks.3.1.bn2.running_var",
             "feature_extractor.bn2.running_var")
    for tag, adapt in (("a", False), ("b", True)):
        model = make_reference_recogniser(fsr, "efficientnet_b0", adapt, "proto", 1, batch_size, num_lite,
                                          film_strength=0.02, learn_extractor=True)
        model.set_test_mode(False)
        model._clear_caches()
        model.zero_grad()
        np.random.seed(900)
        model.personalise_with_lite(ctx, lab)
        logits = model.predict_a_batch(tgt)
        loss = len(lab) / (num_lite * tasks_per_batch) * F.cross_entropy(logits, tlab)
        loss = loss + 0.001 * model.film_generator.regularization_term()
        loss.backward()
        out[tag + "_logits_0"], out[tag + "_loss_0"] = logits, loss
        model._reset()
        params = dict(model.named_parameters())
        for name in watched:
            if name in params and params[name].grad is not None:
                flat = params[name].grad.flatten()
                out["%s_grad__%s" % (tag, name)] = flat[::max(1, flat.numel() // 4096)][:4096].clone()
                out["%s_gnorm__%s" % (tag, name)] = flat.double().norm().float()
        sd = model.state_dict()
        for name in stats:
            out["%s_stat__%s" % (tag, name)] = sd[name].float()
    save("G9_lite_efficientnet", **out)


def g10_heads_versa_mahalanobis():
    """The other two single-step heads, run by the reference's own classes on small feature sets (D = 64 / 96; the
    second case has a single-example class, which takes _estimate_cov's scalar branch)."""
    from model.classifier_heads import MahalanobisClassifier, VersaClassifier
    out = {}
    for tag, D, labels in (("w5", 64, [0] * 6 + [1] * 5 + [2] * 7 + [3] * 4 + [4] * 6),
                           ("single", 96, [3] * 5 + [7] * 1 + [9] * 6)):
        g = torch.Generator().manual_seed(1200 + D)
        lab = torch.tensor(labels)[torch.randperm(len(labels), generator=g)]
        cls = {c: 0.8 * torch.randn(D, generator=g) for c in set(labels)}
        feats = torch.stack([cls[int(c)] for c in lab]) + torch.randn(len(labels), D, generator=g)
        q = torch.stack([cls[int(c)] for c in lab[:9]]) + torch.randn(9, D, generator=g)
        out[tag + "_features"], out[tag + "_labels"], out[tag + "_query"] = feats, lab, q
        versa = VersaClassifier(D, logit_scale=2.0)
        synthetic.init_parameters_(versa, prefix="classifier.")
        with torch.no_grad():
            versa.configure(feats, lab)
            out[tag + "_versa_weight"], out[tag + "_versa_bias"] = versa.weight.detach(), versa.bias.detach()
            out[tag + "_versa_logits"] = versa.predict(q)
        maha = MahalanobisClassifier(logit_scale=1.0)
        with torch.no_grad():
            maha.configure(feats, lab)
            out[tag + "_maha_means"], out[tag + "_maha_precisions"] = maha.means.detach(), maha.precisions.detach()
            out[tag + "_maha_task_precision"] = maha.task_precision.detach()
            out[tag + "_maha_logits"] = maha.predict(q)
    save("G10_heads_versa_mahalanobis", **out)


def g11_finetuner(fsr):
    """MultiStepFewShotRecogniser (the FineTuner baseline): 3 Adam steps on the context set in mini-batches of 4, then
    predict — head only / + FiLM parameters unfrozen / whole extractor unfrozen (few_shot_recognisers.py:185-269)."""
    import torch.nn.functional as F
    task = synthetic.make_task(61, way=3, shots=1, frames_per_shot=3, num_query=6, frame_size=32)
    ctx, lab, tgt = task["context_clips"], task["context_labels"], task["target_clips"]
    out = {"context_clips": ctx, "context_labels": lab, "target_clips": tgt}

    def factory(feature_extractor_name, pretrained, with_film=False, learn_extractor=True):
        fe = oracle_extractors.create(feature_extractor_name)
        synthetic.init_parameters_(fe)
        if not learn_extractor:
            for p in fe.parameters():
                p.requires_grad = False
        names = None
        if with_film:
            mods = dict(fe.named_modules())
            for n in fe.film_slot_names():
                mods[n].film = True
            from model.film import get_film_parameter_names
            names = get_film_parameter_names(feature_extractor_name, fe)
        return fe, names

    fsr.create_feature_extractor = factory
    for tag, adapt, learn in (("head", False, False), ("film", True, False), ("full", False, True)):
        model = fsr.MultiStepFewShotRecogniser("resnet18", adapt, "linear", 1, 4, learn, 1.0)
        model._set_device(torch.device("cpu"))
        model._send_to_device()
        model.set_test_mode(True)
        args = {"num_grad_steps": 3, "learning_rate": 0.01, "extractor_lr_scale": 0.5, "loss_fn": F.cross_entropy,
                "optimizer": "adam", "momentum": 0.0, "weight_decay": 0.0, "betas": (0.9, 0.999), "epsilon": 1e-8}
        model.personalise(ctx, lab, args)
        with torch.no_grad():
            out[tag + "_logits"] = model.predict(tgt)
        out[tag + "_classifier_weight"] = model.classifier.weight.detach()
        out[tag + "_classifier_bias"] = model.classifier.bias.detach()
        sd = model.state_dict()
        out[tag + "_bn1_weight"] = sd["feature_extractor.bn1.weight"]
        out[tag + "_layer4_bn2_bias"] = sd["feature_extractor.layer4.1.bn2.bias"]
        flat = sd["feature_extractor.layer3.0.conv1.weight"].flatten()
        out[tag + "_layer3_conv1_weight"] = flat[::max(1, flat.numel() // 4096)][:4096].clone()
    save("G11_finetuner", **out)


def g13_checkpoint(fsr):
    """§8(f3) checkpoint compatibility (single-step-learner.py:300-305,377-390): what the REFERENCE's model writes
    with state_dict() and what it computes after `load_state_dict(torch.load(path))`.

    The reference model is constructed on "pretrained" extractor values A (the factory initialises the extractor before
    FilmParameterGenerator snapshots gamma0/beta0, as timm's pretrained download does), then loads a checkpoint whose
    extractor equals A (FiLM-replaced BatchNorm parameters never get a gradient) and whose set encoder / FiLM generator
    hold other values. The fixture stores the key list, shapes and dtypes of the reference's state_dict, whether the
    snapshot is part of it, and the logits after the load; checkpoint values are regenerated from (seed, key) by
    synthetic.synth_tensor, so no weights are stored."""
    import tempfile
    out = {}
    task = synthetic.make_task(71, way=3, shots=1, frames_per_shot=3, num_query=7, frame_size=64)
    out["context_clips"], out["context_labels"], out["target_clips"] = (task["context_clips"], task["context_labels"],
                                                                        task["target_clips"])
    for tag, fe_name, adapt in (("effnet_film", "efficientnet_b0", True), ("resnet_proto", "resnet18", False)):
        model = make_reference_recogniser(fsr, fe_name, adapt, "proto", 1, 4, 16,
                                          film_strength=0.02 if fe_name == "efficientnet_b0" else 0.1)
        sd = model.state_dict()
        out[tag + "_keys"] = np.array(list(sd.keys()))
        out[tag + "_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        out[tag + "_dtypes"] = np.array([str(v.dtype) for v in sd.values()])
        out[tag + "_snapshot_in_state_dict"] = np.array(any("initial_film_parameters" in k for k in sd))
        # a checkpoint "trained elsewhere": everything outside the extractor re-drawn with another seed
        ckpt = {k: (v.clone() if k.startswith("feature_extractor.") else
                    synthetic.synth_tensor(k, tuple(v.shape), seed=7, film_strength=0.02)) for k, v in sd.items()}
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "checkpoint.pt")
            torch.save(ckpt, path)
            fresh = make_reference_recogniser(fsr, fe_name, adapt, "proto", 1, 4, 16)
            fresh.load_state_dict(torch.load(path))
        fresh.set_test_mode(True)
        with torch.no_grad():
            fresh.personalise(task["context_clips"], task["context_labels"])
            out[tag + "_logits"] = fresh.predict(task["target_clips"])
            if adapt:
                out[tag + "_film_bn1_weight"] = fresh.film_dict["bn1.weight"]
    save("G13_checkpoint", **out)


# ---- G14: the dataset side (data/datasets.py), f3 -----------------------------------------------------------------------
G14_TREE = {  # user -> object -> (frames per clean video, frames per clutter video); < 50 clutter frames = below the target floor
    "P100": {"mug": ((7, 12, 9, 55), (53, 30)), "keys": ((10, 5), (61,)), "remote": ((8,), (20, 12))},   # remote: no valid target
    "P200": {"cane": ((6, 6, 6, 6, 6, 6, 6), (50, 57)), "wallet": ((11, 4, 52), (70,)), "bottle": ((9, 13, 3), (52, 51, 10))},
    "P300": {"lamp": ((5,), (49,))},                                                                  # user with no valid object
}


def g14_write_tree(root, frame_size=16, seed=1414):
    """root/<user>/<object>/<clean|clutter>/<video>/<video>-NNNNN.jpg, ORBIT's layout (reference data/datasets.py:139-167)."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    files = []
    for user, objs in G14_TREE.items():
        for obj, (clean, clutter) in objs.items():
            base = rng.randint(0, 256, size=(4, 4, 3)).astype(np.float32)
            for kind, counts in (("clean", clean), ("clutter", clutter)):
                for v, n in enumerate(counts):
                    name = "%s--%s--%s--%02d" % (user, obj, kind, v)
                    d = os.path.join(root, user, obj, kind, name)
                    os.makedirs(d, exist_ok=True)
                    for f in range(n):
                        small = np.clip(base + rng.normal(0, 25, base.shape) + 1.5 * f, 0, 255).astype(np.uint8)
                        img = Image.fromarray(small).resize((frame_size, frame_size), Image.BILINEAR)
                        path = os.path.join(d, "%s-%05d.jpg" % (name, f + 1))
                        img.save(path, quality=85)
                        files.append(os.path.relpath(path, root))
    return sorted(files)


def g14_datasets():
    """The reference's own dataset classes on a committed JPEG tree. torchvision is absent offline, so its two transforms that
    data/datasets.py:429-430 calls are stubbed with their documented arithmetic (to_tensor: HWC uint8 -> CHW float32 / 255;
    normalize: (x - mean) / std in float32) - everything else (index, sampling, grouping, shuffling) is the reference's code."""
    import random
    import tempfile
    import shutil

    tvf = types.ModuleType("torchvision.transforms.functional")

    def to_tensor(pic):
        a = torch.from_numpy(np.asarray(pic.convert("RGB") if pic.mode != "RGB" else pic).copy())
        return a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    def normalize(t, mean, std):
        mean = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)

    tvf.to_tensor, tvf.normalize = to_tensor, normalize
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms, tvt.functional = tvt, tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    from data.datasets import UserEpisodicORBITDataset  # the REFERENCE's module

    tmp = tempfile.mkdtemp(prefix="g14_")
    root = os.path.join(tmp, "test")
    files = g14_write_tree(root)
    blob, offsets = [], [0]
    for rel in files:
        b = open(os.path.join(root, rel), "rb").read()
        blob.append(np.frombuffer(b, dtype=np.uint8))
        offsets.append(offsets[-1] + len(b))
    out = {"tree_files": np.array(files), "tree_blob": np.concatenate(blob), "tree_offsets": np.array(offsets, dtype=np.int64)}

    file_no = {f: i for i, f in enumerate(files)}

    def rel(paths):  # frame paths -> row numbers of tree_files
        return np.array([file_no[os.path.relpath(p, root)] for p in np.asarray(paths).reshape(-1)],
                        dtype=np.int32).reshape(np.asarray(paths).shape)

    cases = {  # name -> (class, ctor kwargs, seeds, indices)
        "test_default": (UserEpisodicORBITDataset, dict(way_method="max", object_cap=15, shot_methods=("max", "max"), shots=(5, 2),
                         video_types=("clean", "clutter"), subsample_factor=3, clip_methods=("uniform", "random_200"),
                         clip_length=1, test_mode=True, with_caps=False)),
        "test_T4_max": (UserEpisodicORBITDataset, dict(way_method="max", object_cap=2, shot_methods=("specific", "fixed"), shots=(2, 1),
                        video_types=("clean", "clutter"), subsample_factor=2, clip_methods=("max", "max"),
                        clip_length=4, test_mode=True, with_caps=False)),
        "train_random": (UserEpisodicORBITDataset, dict(way_method="random", object_cap=15, shot_methods=("random", "random"), shots=(5, 2),
                         video_types=("clean", "clutter"), subsample_factor=4, clip_methods=("uniform", "random"),
                         clip_length=1, test_mode=False, with_caps=True)),
        "train_cleanclean_T3": (UserEpisodicORBITDataset, dict(way_method="max", object_cap=15, shot_methods=("fixed", "max"), shots=(3, 2),
                                video_types=("clean", "clean"), subsample_factor=1, clip_methods=("max", "max"),
                                clip_length=3, test_mode=False, with_caps=False)),
        # (ObjectEpisodicORBITDataset.__getitem__ calls sample_task without its task_id argument, reference :637: a TypeError
        # as written, so there is nothing of it to record)
        "train_r200_uniform": (UserEpisodicORBITDataset, dict(way_method="random", object_cap=2, shot_methods=("max", "fixed"), shots=(5, 1),
                               video_types=("clean", "clutter"), subsample_factor=5, clip_methods=("random_200", "uniform"),
                               clip_length=1, test_mode=False, with_caps=False)),
    }
    for name, (cls, kw) in cases.items():
        ds = cls(root, kw["way_method"], kw["object_cap"], kw["shot_methods"], kw["shots"], kw["video_types"],
                 kw["subsample_factor"], kw["clip_methods"], kw["clip_length"], 16, "imagenet", [], ([], []), kw["test_mode"],
                 False, kw["with_caps"], None)
        out[name + "_users"] = np.array(ds.users)
        out[name + "_num_objects"] = np.array(ds.num_objects)
        out[name + "_video_ids"] = np.array([os.path.relpath(p, root) for p, _ in sorted(ds.video2id.items(), key=lambda kv: kv[1])])
        out[name + "_video_frames"] = np.array([len(ds.vid2frames[p]) for p, _ in sorted(ds.video2id.items(), key=lambda kv: kv[1])])
        random.seed(1991 + len(name))
        n_items = len(ds)
        for rep in range(2):          # two passes: the module-level random stream runs on (with_caps state persists)
            for idx in range(n_items):
                t = ds[idx]
                key = "%s_r%d_i%d" % (name, rep, idx)
                out[key + "_objects"] = np.array(t["object_list"])
                out[key + "_context_paths"] = rel(t["context_paths"])
                out[key + "_context_labels"] = t["context_labels"]
                if rep == 0 and idx == 0:
                    out[key + "_context_clips"] = t["context_clips"]
                out[key + "_context_clips_sum"] = t["context_clips"].double().sum(dim=(1, 2, 3, 4))
                if kw["test_mode"]:
                    out[key + "_target_videos"] = np.array(len(t["target_paths"]))
                    for v, (fr, pa, la) in enumerate(zip(t["target_clips"], t["target_paths"], t["target_labels"])):
                        out[key + "_target%d_paths" % v] = rel(pa)
                        out[key + "_target%d_label" % v] = la
                        out[key + "_target%d_frames_sum" % v] = fr.double().sum(dim=(1, 2, 3))
                        if v == 0 and rep == 0 and idx == 0:
                            out[key + "_target0_frames"] = fr
                else:
                    out[key + "_target_paths"] = rel(t["target_paths"])
                    out[key + "_target_labels"] = t["target_labels"]
                    out[key + "_target_clips_sum"] = t["target_clips"].double().sum(dim=(1, 2, 3, 4))
    out["case_names"] = np.array(list(cases))
    shutil.rmtree(tmp)
    save("G14_datasets", **out)


def g7_utils():
    from data.utils import attach_frame_history, get_batch_indices
    frames = torch.arange(6 * 3 * 2 * 2, dtype=torch.float32).reshape(6, 3, 2, 2)
    save("G7_utils", frames=frames, hist1=attach_frame_history(frames, 1), hist3=attach_frame_history(frames, 3),
         batch_10_4=np.array([get_batch_indices(i, 10, 4) for i in range(3)]),
         batch_257_256=np.array([get_batch_indices(i, 257, 256) for i in range(2)]))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    stub_timm()
    only = [a for a in sys.argv[2:]]
    if only:  # e.g. `make_golden.py /root/reference g13_checkpoint`: regenerate selected fixtures only
        import model.few_shot_recognisers as fsr
        for name in only:
            fn = globals()[name]
            fn(fsr) if fn.__code__.co_argcount else fn()
        return
    g1_head()
    g2_pooler()
    g3_set_encoder()
    g4_film_generator()
    g7_utils()
    g10_heads_versa_mahalanobis()
    import model.few_shot_recognisers as fsr
    g5_recogniser(fsr)
    g6_lite(fsr)
    g8_lite_learn_extractor(fsr)
    g9_lite_efficientnet(fsr)
    g11_finetuner(fsr)
    g13_checkpoint(fsr)
    g14_datasets()


if __name__ == "__main__":
    main()
