"""Meta-training on the HIP path: native tapes + backward kernels behind torch.autograd.

* G6 / G8 golden fixtures: gradients, losses, logits and BatchNorm running statistics recorded from the REFERENCE's
  own `personalise_with_lite` / `predict_a_batch` / `loss.backward()` (tests/golden/make_golden.py) — frozen
  extractor + FiLM (G6), unfrozen extractor with train-mode BatchNorm (G8a), both (G8b).
* extractor / set-encoder level: orbit_extractor_train_forward + orbit_extractor_backward against torch autograd on
  the CPU oracle modules at other sizes and batch shapes.

Tolerances. The kernels themselves are fp32-exact: against the fp64 oracle every parameter gradient of resnet18 is
within ~3e-6 of its largest magnitude. What is NOT stable across implementations is the ReLU mask: a pre-activation
within rounding distance of zero takes the other branch on the GPU than on the CPU, which perturbs every upstream
gradient by 1e-3..3e-2. Round 1 tolerated that with loose bounds; now the masks are ALIGNED (tests/relu_align.py): when the
plain comparison misses the exact bound, the oracle's ReLU units with |pre-activation| < 2e-5 are flipped (at most three,
chosen greedily) and the oracle must then reproduce the GPU's gradients to the exact bound - 2e-5 against the fp64
oracle for EVERY seed, 1e-3 for the LITE flows against the fp32 oracle (itself within 1e-3 of the gradients the
reference recorded, tests/test_oracle_golden.py). Logits keep the 1e-3 absolute bar, running statistics 1e-4.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import orbit_dataset_amd  # noqa: E402,F401
from relu_align import ReluTap, aligned_error  # noqa: E402
from oracle import blocks as oracle_blocks  # noqa: E402
from oracle import extractors as oracle_extractors  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402
from orbit_dataset_amd.model.film import get_film_parameters  # noqa: E402
from orbit_dataset_amd.model.set_encoders import SetEncoder  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return {k: (torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v)
            for k, v in np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False).items()}


def native(adapt, batch_size, num_lite, learn_extractor, fe_name="resnet18", film_strength=0.1):
    m = SingleStepFewShotRecogniser(fe_name, adapt, "proto", 1, batch_size, learn_extractor, num_lite, 1.0)
    synthetic.init_parameters_(m, film_strength=film_strength)
    if adapt:
        m.film_generator.initial_film_parameters = get_film_parameters(m.film_parameter_names, m.feature_extractor)
    m._set_device("cuda:0")
    m._send_to_device()
    return m


def rel(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def lite_steps(m, g, seed0, prefix="", batches=2):
    """Learner.train_task_with_lite's loop (single-step-learner.py:212-243) over the query batches."""
    nl, tpb, bs = int(g["num_lite_samples"]), int(g["tasks_per_batch"]), int(g["batch_size"])
    dev = torch.device("cuda:0")
    ctx, lab = g["context_clips"].to(dev), g["context_labels"].to(dev)
    m.set_test_mode(False)
    m._clear_caches()
    m.zero_grad()
    for b in range(batches):
        np.random.seed(seed0 + b)
        m.personalise_with_lite(ctx, lab)
        logits = m.predict_a_batch(g["target_clips"][b * bs:(b + 1) * bs].to(dev))
        want = g[prefix + "logits_%d" % b]
        assert (logits.detach().cpu() - want).abs().max().item() < 1e-3, "logits of batch %d" % b
        loss = len(lab) / (nl * tpb) * F.cross_entropy(logits, g["target_labels"][b * bs:(b + 1) * bs].to(dev))
        loss = loss + 0.001 * m.film_generator.regularization_term()
        assert abs(loss.item() - float(g[prefix + "loss_%d" % b])) < 1e-3
        loss.backward()
        m._reset()


def _oracle_lite(g, adapt, learn_extractor, seeds):
    """The same LITE steps on the fp32 CPU oracle (oracle/training.py, pinned to the reference's recorded gradients at
    1e-3 by tests/test_oracle_golden.py) with its ReLUs instrumented. Returns (run, tap); run() restores the initial
    parameters / running statistics first (train-mode BatchNorm mutates them) and returns {name: gradient}."""
    from oracle.recogniser import OracleRecogniser
    from oracle.training import LiteTrainer
    nl, tpb, bs = int(g["num_lite_samples"]), int(g["tasks_per_batch"]), int(g["batch_size"])
    ref = OracleRecogniser("resnet18", adapt, "proto", 1, bs, nl)
    synthetic.init_parameters_(ref.fe)
    mods = {"feature_extractor.": ref.fe}
    if adapt:
        synthetic.init_parameters_(ref.set_encoder)
        synthetic.init_parameters_(ref.build_film_generator(), prefix="film_generator.")
        mods.update({"set_encoder.": ref.set_encoder, "film_generator.": ref.film_generator})
    tr = LiteTrainer(ref, learn_extractor, tpb)
    init = {pre: {k: v.clone() for k, v in mod.state_dict().items()} for pre, mod in mods.items()}
    tap = ReluTap(ref.fe, ref.set_encoder)

    def run():
        for pre, mod in mods.items():
            mod.load_state_dict(init[pre])
            for prm in mod.parameters():
                prm.grad = None
        n = 2 * bs
        tr.train_task_with_lite(g["context_clips"], g["context_labels"], g["target_clips"][:n], g["target_labels"][:n],
                                seeds=seeds)
        return {pre + k: prm.grad for pre, mod in mods.items() for k, prm in mod.named_parameters()
                if prm.grad is not None}
    return run, tap


def _check_lite_gradients(m, g, adapt, learn_extractor, seeds, tag=""):
    """GPU gradients of ALL parameters against the fp32 oracle at 1e-3 (ReLU masks aligned if needed), and the sampled
    gradients the reference recorded: directly at 1e-3 when no mask flipped."""
    run, tap = _oracle_lite(g, adapt, learn_extractor, seeds)
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}

    def error_of(grads):
        assert sorted(grads) == sorted(got)
        return max(rel(got[n], grads[n]) for n in grads)

    err, flipped, plain = aligned_error(run, error_of, tap, exact=1e-3)
    assert err < 1e-3, "GPU vs oracle gradients %g (plain %g, flipped %s)" % (err, plain, flipped)
    prefix = tag + "_grad__" if tag else "grad__"
    checked = 0
    for key in g:
        if not key.startswith(prefix):
            continue
        name = key[len(prefix):]
        flat = got[name].flatten()
        sample = flat[::max(1, flat.numel() // 4096)][:4096] if tag else got[name]  # G6 stores whole gradients
        if not flipped:  # same masks as the reference's run: the recorded gradients must match directly
            assert rel(sample, g[key]) < 1e-3, (name, rel(sample, g[key]))
        checked += 1
    return checked, flipped


def test_G6_lite_backward_frozen_extractor(device):
    g = gold("G6_lite")
    m = native(True, int(g["batch_size"]), int(g["num_lite_samples"]), False)
    lite_steps(m, g, 500)
    checked, _ = _check_lite_gradients(m, g, True, False, (500, 501))
    assert checked >= 4
    assert not bool(g["extractor_has_grad"])
    assert all(p.grad is None for p in m.feature_extractor.parameters())


@pytest.mark.parametrize("tag,adapt", [("a", False), ("b", True)])
def test_G8_lite_unfrozen_extractor(device, tag, adapt):
    g = gold("G8_lite_learn_extractor")
    m = native(adapt, int(g["batch_size"]), int(g["num_lite_samples"]), True)
    lite_steps(m, g, 800, prefix=tag + "_")
    params = dict(m.named_parameters())
    checked, flipped = _check_lite_gradients(m, g, adapt, True, (800, 801), tag=tag)
    assert checked >= 6
    if not flipped:
        for key in g:
            if key.startswith(tag + "_gnorm__"):
                name = key[len(tag + "_gnorm__"):]
                gn = float(g[key])
                assert abs(float(params[name].grad.flatten().double().norm()) - gn) < 1e-3 * gn, name
    sd = m.state_dict()
    for key in g:
        if key.startswith(tag + "_stat__"):
            name = key[len(tag + "_stat__"):]
            want = g[key] if isinstance(g[key], torch.Tensor) else torch.tensor(float(g[key]))
            assert rel(sd[name].float(), want) < 1e-4, name
    has = params["feature_extractor.bn1.weight"].grad is not None
    assert has == bool(g[tag + "_bn1_weight_has_grad"])


@pytest.mark.parametrize("query_too", [False, True], ids=["subset", "subset+query"])
def test_lite_subset_pass_beside_cache_pass_changes_nothing(device, query_too, monkeypatch):
    """The first query batch of a LITE task re-encodes the H-clip subset on a second stream beside the cache pass, with the
    subset's running-statistics update deferred (ORBIT_TRAIN_DEFER_RUNNING_STATS + orbit_extractor_apply_deferred_bn_stats);
    with `lite_query_overlap` the query batch's taped pass starts from the same fork point on a third stream, recording into
    buffers the network keeps. Logits and every gradient are bit-identical to the serial order, the running statistics agree
    to the last bit or two (the deferred update evaluates the same expression in another kernel)."""
    g = gold("G8_lite_learn_extractor")
    outs = []
    # every tape starts as 0xFF bytes (NaN floats): the deferred statistics must not depend on what the slot held before
    # (a momentum-1 update written as 0 * old + 1 * new poisoned the running statistics in ~4 % of the meta-training runs)
    from orbit_dataset_amd.model import autograd as native_autograd
    plain_empty = native_autograd._empty_bytes
    monkeypatch.setattr(native_autograd, "_empty_bytes", lambda n, dev: plain_empty(n, dev).fill_(255))
    for overlap in (True, False):
        m = native(False, int(g["batch_size"]), int(g["num_lite_samples"]), True)
        m.lite_overlap = overlap
        m.lite_query_overlap = overlap and query_too
        lite_steps(m, g, 800, prefix="a_")
        torch.cuda.synchronize()
        outs.append(({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                     {k: v.detach().clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}))
    (g1, s1), (g0, s0) = outs
    assert sorted(g1) == sorted(g0) and len(g1) > 20
    for n in g1:
        assert torch.equal(g1[n], g0[n]), n
    assert sorted(s1) == sorted(s0) and len(s1) > 20
    for k in s1:
        if "num_batches" in k:
            assert torch.equal(s1[k], s0[k]), k
        else:
            assert rel(s1[k].float().cpu(), s0[k].float().cpu()) < 1e-6, k


@pytest.mark.parametrize("tag,adapt", [("a", False), ("b", True)])
def test_G9_lite_efficientnet(device, tag, adapt):
    """The README's main recipe on the efficientnet_b0 plan: gradients recorded from the reference's loss.backward().
    Smooth activations only (SiLU / sigmoid): no mask flips, so the bar is 1e-3 on every sampled gradient."""
    g = gold("G9_lite_efficientnet")
    m = native(adapt, int(g["batch_size"]), int(g["num_lite_samples"]), True, "efficientnet_b0", 0.02)
    lite_steps(m, g, 900, prefix=tag + "_", batches=1)
    params = dict(m.named_parameters())
    top = max(float(g[k].abs().max()) for k in g if k.startswith(tag + "_grad__"))
    checked = 0
    for key in g:
        if not key.startswith(tag + "_grad__"):
            continue
        name = key[len(tag + "_grad__"):]
        flat = params[name].grad.flatten()
        sample = flat[::max(1, flat.numel() // 4096)][:4096].cpu()
        err = float((sample.double() - g[key].double()).abs().max()) / max(float(g[key].abs().max()), 1e-4 * top)
        assert err < 1e-3, (name, err)
        checked += 1
    assert checked >= 10
    sd = m.state_dict()
    for key in g:
        if key.startswith(tag + "_stat__"):
            assert rel(sd[key[len(tag + "_stat__"):]].float(), g[key]) < 1e-4, key


@pytest.mark.parametrize("tag,adapt,learn", [("head", False, False), ("film", True, False), ("full", False, True)])
def test_G11_finetuner(device, tag, adapt, learn):
    """MultiStepFewShotRecogniser (FineTuner) on the native backward kernels against the reference's run: 3 Adam steps
    over mini-batches of 4 context clips — linear head only / + FiLM BatchNorm parameters / whole extractor."""
    from orbit_dataset_amd.model.few_shot_recognisers import MultiStepFewShotRecogniser
    from orbit_dataset_amd.optim import cross_entropy
    g = gold("G11_finetuner")
    m = MultiStepFewShotRecogniser("resnet18", adapt, "linear", 1, 4, learn, 1.0)
    synthetic.init_parameters_(m)
    m._set_device("cuda:0")
    m._send_to_device()
    m.set_test_mode(True)
    args = {"num_grad_steps": 3, "learning_rate": 0.01, "extractor_lr_scale": 0.5, "loss_fn": cross_entropy,
            "optimizer": "adam", "momentum": 0.0, "weight_decay": 0.0, "betas": (0.9, 0.999), "epsilon": 1e-8}
    m.personalise(g["context_clips"].to(device), g["context_labels"].to(device), args)
    with torch.no_grad():
        logits = m.predict(g["target_clips"].to(device)).cpu()
    want = g[tag + "_logits"]
    # Adam normalises every gradient by its own magnitude, so elements whose gradient is rounding noise move by
    # +-lr per step in an implementation-dependent direction: parameters are compared at 3 steps x lr = 3e-2 worst case,
    # the logits (which average over 512 features) tightly
    assert (logits - want).abs().max().item() < 2e-2 * max(1.0, want.abs().max().item())
    assert torch.equal(logits.argmax(1), want.argmax(1))
    sd = m.state_dict()
    assert (sd["classifier.weight"].cpu() - g[tag + "_classifier_weight"]).abs().max().item() < 3.1e-2
    frozen = not (adapt or learn)
    tol = 0.0 if frozen else 1.6e-2
    assert (sd["feature_extractor.bn1.weight"].cpu() - g[tag + "_bn1_weight"]).abs().max().item() <= tol
    if not learn:  # filters untouched unless the whole extractor is unfrozen
        flat = sd["feature_extractor.layer3.0.conv1.weight"].flatten().cpu()
        assert torch.equal(flat[::max(1, flat.numel() // 4096)][:4096], g[tag + "_layer3_conv1_weight"])
    m._reset()


# ---- extractor level, against torch autograd on the oracle modules ---------------------------------------------------
def _oracle_and_native(name, device, requires_grad=True):
    if name == "set_encoder":
        ref, nat = oracle_blocks.SetEncoder(), SetEncoder()
    else:
        ref = oracle_extractors.create(name)
        nat, _ = create_feature_extractor(name, with_film=True, learn_extractor=requires_grad)
    synthetic.init_parameters_(ref)
    synthetic.init_parameters_(nat)
    return ref, nat.to(device)


def _seeded_check(run, tap, seeds=5, exact=2e-5):
    """run(seed) -> (run_oracle, error_of): EVERY seed must reach the fp32-exact bound, directly or - when a ReLU mask
    flipped between CPU and GPU - after flipping at most three fragile units of the oracle (tests/relu_align.py)."""
    report = []
    for seed in range(seeds):
        run_oracle, error_of = run(seed)
        err, flipped, plain = aligned_error(run_oracle, error_of, tap, exact)
        report.append((seed, plain, err, flipped))
        assert err < exact, "seed %d: %g after aligning %s (plain %g)" % (seed, err, flipped, plain)
    return report


@pytest.mark.parametrize("size,B,bn_train", [(64, 6, True), (33, 4, True), (33, 4, False)])
def test_resnet18_backward_matches_autograd(device, bn_train, size, B):
    ref, nat = _oracle_and_native("resnet18", device)
    ref = ref.double()
    tap = ReluTap(ref)
    sd, rsd = {k: v.clone() for k, v in nat.state_dict().items()}, {k: v.clone() for k, v in ref.state_dict().items()}

    def run(seed):
        nat.load_state_dict(sd)
        nat.zero_grad()
        nat.train(bn_train)
        x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(100 * size + seed))
        dfeat = torch.randn(B, 512, generator=torch.Generator().manual_seed(seed))
        out = nat(x.to(device))
        out.backward(dfeat.to(device))
        got = {name: p.grad.detach().cpu() for name, p in nat.named_parameters()}
        state = {}

        def run_oracle():
            ref.load_state_dict(rsd)
            ref.zero_grad()
            ref.train(bn_train)
            state["out"] = ref(x.double())
            state["out"].backward(dfeat.double())
            return {name: p.grad for name, p in ref.named_parameters()}

        def error_of(grads):
            return max(rel(got[name], grads[name]) for name in got)

        run_oracle()
        assert rel(out.detach(), state["out"].detach()) < 2e-5
        if bn_train:
            ref_sd = ref.state_dict()
            for name, buf in nat.state_dict().items():
                if "running" in name:
                    assert rel(buf, ref_sd[name]) < 1e-5, name
                elif "num_batches_tracked" in name:
                    assert int(buf) == int(ref_sd[name]) == 1
        return run_oracle, error_of

    _seeded_check(run, tap)


def test_resnet18_film_gradients_frozen_extractor(device):
    """CNAPs-style: frozen extractor in eval mode, gradients only w.r.t. the per-task FiLM vectors."""
    from torch.func import functional_call
    ref, nat = _oracle_and_native("resnet18", device, requires_grad=False)
    ref = ref.double().eval()
    nat.eval()
    for p in ref.parameters():
        p.requires_grad = False
    slots = [n for n, _ in nat.film_slot_modules()]
    params = dict(ref.named_parameters())

    tap = ReluTap(ref)

    def run(seed):
        gen = torch.Generator().manual_seed(11 + seed)
        film_vals, gam, bet = {}, [], []
        for n in slots:
            w0, b0 = params[n + ".weight"].detach(), params[n + ".bias"].detach()
            gvec = w0 * (1 + 0.1 * torch.randn(w0.shape, generator=gen, dtype=torch.float64))
            bvec = b0 + 0.1 * torch.randn(b0.shape, generator=gen, dtype=torch.float64)
            film_vals[n + ".weight"], film_vals[n + ".bias"] = gvec, bvec
            gam.append(gvec.float()), bet.append(bvec.float())
        x = torch.randn(5, 3, 64, 64, generator=gen)
        dfeat = torch.randn(5, 512, generator=gen)
        gamma = torch.cat(gam).to(device).requires_grad_(True)
        beta = torch.cat(bet).to(device).requires_grad_(True)
        nat(x.to(device), film=(gamma, beta)).backward(dfeat.to(device))
        assert all(p.grad is None for p in nat.parameters())
        got = (gamma.grad.cpu(), beta.grad.cpu())

        def run_oracle():
            film_ref = {k: v.clone().requires_grad_(True) for k, v in film_vals.items()}
            functional_call(ref, film_ref, (x.double(),)).backward(dfeat.double())
            return (torch.cat([film_ref[n + ".weight"].grad for n in slots]),
                    torch.cat([film_ref[n + ".bias"].grad for n in slots]))

        def error_of(grads):
            return max(rel(got[0], grads[0]), rel(got[1], grads[1]))

        return run_oracle, error_of

    _seeded_check(run, tap)


@pytest.mark.parametrize("size,B", [(32, 3), (84, 5), (50, 2)])
def test_set_encoder_backward_matches_autograd(device, size, B):
    ref, nat = _oracle_and_native("set_encoder", device)
    ref = ref.double().eval()
    nat.eval()  # the set encoder always normalises with running statistics (few_shot_recognisers.py:176-183)

    tap = ReluTap(ref)

    def run(seed):
        nat.zero_grad()
        x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(size + seed))
        dfeat = torch.randn(B, 64, generator=torch.Generator().manual_seed(2 + seed))
        out = nat(x.to(device))
        out.backward(dfeat.to(device))
        got = {name: p.grad.detach().cpu() for name, p in nat.named_parameters()}
        state = {}

        def run_oracle():
            ref.zero_grad()
            state["out"] = ref(x.double())
            state["out"].backward(dfeat.double())
            return {name: p.grad for name, p in ref.named_parameters()}

        def error_of(grads):
            return max(rel(got[name], grads[name]) for name in got)

        run_oracle()
        assert rel(out.detach(), state["out"].detach()) < 2e-5
        return run_oracle, error_of

    _seeded_check(run, tap)


def test_backward_is_deterministic(device):
    _, nat = _oracle_and_native("resnet18", device)
    nat.train()
    x = torch.randn(8, 3, 64, 64, generator=torch.Generator().manual_seed(4)).to(device)
    dfeat = torch.randn(8, 512, generator=torch.Generator().manual_seed(5)).to(device)
    sd0 = {k: v.clone() for k, v in nat.state_dict().items()}
    grads = []
    for _ in range(2):
        nat.load_state_dict(sd0)
        nat.zero_grad()
        nat(x).backward(dfeat)
        grads.append(torch.cat([p.grad.flatten() for p in nat.parameters()]).cpu())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("size,B,bn_train", [(64, 4, True), (96, 3, False)])
def test_efficientnet_backward_matches_autograd(device, bn_train, size, B):
    """Depthwise / squeeze-excite / SiLU backward through the whole tf_efficientnet_b0 plan. No ReLU here, so there
    are no mask flips: every seed has to be close to fp32-exact (the fast exp/rcp of SiLU and sigmoid costs a little)."""
    ref, nat = _oracle_and_native("efficientnet_b0", device)
    ref = ref.double()
    sd, rsd = {k: v.clone() for k, v in nat.state_dict().items()}, {k: v.clone() for k, v in ref.state_dict().items()}

    def run(seed):
        nat.load_state_dict(sd), ref.load_state_dict(rsd)
        nat.zero_grad(), ref.zero_grad()
        ref.train(bn_train), nat.train(bn_train)
        x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(7 * size + seed))
        dfeat = torch.randn(B, 1280, generator=torch.Generator().manual_seed(seed))
        out_ref = ref(x.double())
        out_ref.backward(dfeat.double())
        out = nat(x.to(device))
        assert rel(out.detach(), out_ref.detach()) < 5e-5
        out.backward(dfeat.to(device))
        ref_grads = dict(ref.named_parameters())
        # some gradients vanish in exact arithmetic (a bias in front of a 1x1 conv + batch-statistics BatchNorm): those
        # (reference magnitude < 1e-5 of the largest gradient in the net) must come out as rounding noise of that scale;
        # every other tensor is measured against its own largest magnitude
        top = max(float(g.grad.abs().max()) for g in ref_grads.values())
        errs = {}
        for name, p in nat.named_parameters():
            want = ref_grads[name].grad
            diff = float((p.grad.double().cpu() - want).abs().max())
            mag = float(want.abs().max())
            errs[name] = diff / mag if mag > 1e-5 * top else 2e-4 * diff / (1e-5 * top)
        if bn_train:
            ref_sd = ref.state_dict()
            for name, buf in nat.state_dict().items():
                if "running" in name:
                    assert rel(buf, ref_sd[name]) < 2e-5, name
        worst = max(errs, key=errs.get)
        assert errs[worst] < 2e-4, (worst, errs[worst])
        return errs[worst]

    run(0)


def test_efficientnet_film_gradients_frozen_extractor(device):
    from torch.func import functional_call
    ref, nat = _oracle_and_native("efficientnet_b0", device, requires_grad=False)
    ref = ref.double().eval()
    nat.eval()
    for p in ref.parameters():
        p.requires_grad = False
    slots = [n for n, _ in nat.film_slot_modules()]
    params = dict(ref.named_parameters())
    gen = torch.Generator().manual_seed(3)
    film_ref, gam, bet = {}, [], []
    for n in slots:
        w0, b0 = params[n + ".weight"].detach(), params[n + ".bias"].detach()
        gvec = w0 * (1 + 0.02 * torch.randn(w0.shape, generator=gen, dtype=torch.float64))
        bvec = b0 + 0.02 * torch.randn(b0.shape, generator=gen, dtype=torch.float64)
        film_ref[n + ".weight"], film_ref[n + ".bias"] = gvec.requires_grad_(True), bvec.requires_grad_(True)
        gam.append(gvec.detach().float()), bet.append(bvec.detach().float())
    x = torch.randn(2, 3, 128, 128, generator=gen)
    dfeat = torch.randn(2, 1280, generator=gen)
    functional_call(ref, film_ref, (x.double(),)).backward(dfeat.double())
    gamma = torch.cat(gam).to(device).requires_grad_(True)
    beta = torch.cat(bet).to(device).requires_grad_(True)
    nat(x.to(device), film=(gamma, beta)).backward(dfeat.to(device))
    dg_ref = torch.cat([film_ref[n + ".weight"].grad for n in slots])
    db_ref = torch.cat([film_ref[n + ".bias"].grad for n in slots])
    assert rel(gamma.grad, dg_ref) < 2e-4 and rel(beta.grad, db_ref) < 2e-4
    assert all(p.grad is None for p in nat.parameters())


def test_training_graph_replay_is_bit_identical(device, lib):
    """Option train_graph = 1: orbit_extractor_train_forward / _backward replay captured HIP graphs once a call repeats an
    earlier call's pointers (same tensors here). Gradients, features and running statistics must equal the eager run bit
    for bit, also after the parameters changed in between (the graph reads the plan's current parameter pool)."""
    _, nat = _oracle_and_native("resnet18", device)
    nat.train()
    x = torch.randn(6, 3, 64, 64, generator=torch.Generator().manual_seed(4)).to(device)
    dfeat = torch.randn(6, 512, generator=torch.Generator().manual_seed(5)).to(device)
    sd0 = {k: v.clone() for k, v in nat.state_dict().items()}

    def run(scale):
        nat.load_state_dict(sd0)
        with torch.no_grad():
            dict(nat.named_parameters())["layer3.0.conv1.weight"].mul_(scale)
        nat.zero_grad()
        out = nat(x)
        out.backward(dfeat)
        return (out.detach().clone(), {n: p.grad.clone() for n, p in nat.named_parameters()},
                {k: v.clone() for k, v in nat.state_dict().items() if "running" in k})

    prev = lib.orbit_get_option(b"train_graph")
    try:
        lib.orbit_set_option(b"train_graph", 0)
        want = {s_: run(s_) for s_ in (1.0, 1.25)}
        lib.orbit_set_option(b"train_graph", 1)
        before = nat.train_graph_stats()
        for rep in range(8):  # sight 1 eager, sight 2 captures, then replays; parameters alternate between the runs
            for s_ in (1.0, 1.25):
                out, grads, stats = run(s_)
                assert torch.equal(out, want[s_][0])
                assert all(torch.equal(grads[n], want[s_][1][n]) for n in grads), (rep, s_)
                assert all(torch.equal(stats[k], want[s_][2][k]) for k in stats)
        after = nat.train_graph_stats()
    finally:
        lib.orbit_set_option(b"train_graph", prev)
    # torch's caching allocator returns the tape / gradient buffers at the same addresses in this steady loop (how often depends
    # on what earlier tests left in its pools: some of the 32 calls must have replayed)
    assert after[0] - before[0] >= 2, (before, after)


@pytest.mark.parametrize("B,size", [(6, 96), (3, 224)])
def test_no_backward_forward_is_bit_identical_to_the_taped_forward(device, B, size):
    """ORBIT_TRAIN_NO_BACKWARD (the batch-statistics forwards LITE issues under torch.no_grad(): few_shot_recognisers.py:
    134-146 cache passes): the depthwise convs apply the preceding BatchNorm + SiLU while loading the raw conv output
    instead of reading a materialised activation. Features AND every running statistic must equal the taped forward's bit
    for bit (same kernels up to where the activation is applied)."""
    from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
    fe, _ = create_feature_extractor("efficientnet_b0", True, False, True)
    synthetic.init_parameters_(fe)
    fe = fe.cuda().train()
    sd0 = {k: v.clone() for k, v in fe.state_dict().items()}
    x = torch.randn(B, 3, size, size, device=device, generator=torch.Generator(device=device).manual_seed(5))
    with torch.enable_grad():
        taped = fe(x).detach().clone()
    sd_taped = {k: v.clone() for k, v in fe.state_dict().items()}
    from orbit_dataset_amd import _lib
    lib = _lib.load()
    fronts = lib.orbit_get_option(b"train_fused_fronts")
    lib.orbit_set_option(b"train_fused_fronts", 0)  # (the two-sweep fused fronts have a test of their own below)
    try:
        fe.load_state_dict(sd0)
        with torch.no_grad():
            plain = fe(x).clone()
        sd_plain = fe.state_dict()
        assert torch.isfinite(taped).all() and torch.equal(plain, taped)
        moved = 0
        for k in sd_taped:
            assert torch.equal(sd_plain[k], sd_taped[k]), k
            moved += int(k.endswith("running_mean") and not torch.equal(sd_taped[k], sd0[k]))
        assert moved >= 40  # the forwards really ran on batch statistics and updated them
        lib.orbit_set_option(b"train_dw_xf", 0)
        try:
            fe.load_state_dict(sd0)
            with torch.no_grad():
                assert torch.equal(fe(x), taped)
        finally:
            lib.orbit_set_option(b"train_dw_xf", 1)
    finally:
        lib.orbit_set_option(b"train_fused_fronts", fronts)


@pytest.mark.parametrize("B,size", [(5, 224), (4, 160), (3, 97)])
def test_no_backward_forward_on_two_sweep_fused_fronts(device, B, size):
    """Round 6 (option train_fused_fronts): on a no-grad batch-statistics forward the expansion conv + depthwise conv of an
    MBConv block run as a STATISTICS SWEEP of the expansion conv (nothing stored) + the row-streaming fused front in its RAW form.
    Against the unfused pair: the first BatchNorm of every fused block sees the SAME statistics bit for bit (same conv kernel,
    same tiles - checked on its running statistics), everything downstream agrees to summation order of the second BatchNorm's
    statistics (features and running statistics to 1e-5 relative). 224: the exact-tiling instantiations; 160 / 97: the guarded
    ones. Option 2 = every shape the fused front serves (the default, 1, takes the 112x112 / 56x56 blocks); 3 = as 2 with the conv's
    own statistics sweep instead of the Gram-matrix statistics."""
    from orbit_dataset_amd import _lib
    from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
    lib = _lib.load()
    fe, _ = create_feature_extractor("efficientnet_b0", True, False, True)
    synthetic.init_parameters_(fe)
    fe = fe.cuda().train()
    sd0 = {k: v.clone() for k, v in fe.state_dict().items()}
    x = torch.randn(B, 3, size, size, device=device, generator=torch.Generator(device=device).manual_seed(6))
    prev = lib.orbit_get_option(b"train_fused_fronts")
    outs = {}
    try:
        for opt in (0, 1, 2, 3):
            lib.orbit_set_option(b"train_fused_fronts", opt)
            fe.load_state_dict(sd0)
            with torch.no_grad():
                feats = fe(x).clone()
            outs[opt] = (feats, {k: v.clone() for k, v in fe.state_dict().items()})
    finally:
        lib.orbit_set_option(b"train_fused_fronts", prev)
    base, sd_base = outs[0]
    assert torch.isfinite(base).all()
    for opt in (1, 2, 3):
        got, sd = outs[opt]
        assert (got - base).abs().max().item() <= 1e-5 * max(1.0, base.abs().max().item()), opt
        for k, v in sd.items():
            if not k.endswith(("running_mean", "running_var")):
                continue
            w = sd_base[k]
            assert (v - w).abs().max().item() <= 1e-5 * max(1.0, w.abs().max().item()), (opt, k)
        # block 1.0 is the first fused front: its bn1 = the expansion conv's BatchNorm sees the same input as on the unfused path
        # (later blocks' inputs already differ in their last bits). Option 3 takes its statistics from a statistics sweep of the
        # conv itself - the same kernel, tiles and summation order as the storing conv: bit-identical; options 1 / 2 from the
        # Gram matrix of the block input (mean = w . mean(x), E[y^2] = w^T E[x x^T] w in double): equal to rounding
        for stat in ("running_mean", "running_var"):
            a, b_ = sd["blocks.1.0.bn1." + stat], sd_base["blocks.1.0.bn1." + stat]
            if opt == 3:
                assert torch.equal(a, b_), stat
            else:
                assert (a - b_).abs().max().item() <= 2e-6 * max(1.0, b_.abs().max().item()), (opt, stat)
    # option 2 really changed something downstream of the first fused block (the second BatchNorm's statistics are summed in
    # another order), i.e. the fused path ran
    assert not torch.equal(outs[2][1]["blocks.1.0.bn2.running_var"], sd_base["blocks.1.0.bn2.running_var"]) or \
        not torch.equal(outs[2][0], base)


@pytest.mark.parametrize("Cin,C,P", [(16, 96, 50_000), (24, 144, 31_337), (24, 144, 9_999), (16, 96, 257)])
def test_bn_statistics_of_a_pointwise_conv_from_the_gram_matrix_of_its_input(device, Cin, C, P):
    """launch_bn_stats_from_gram (round 6): batch mean and 1 / sqrt(biased variance + eps) of y = W x WITHOUT forming y - mean(y_c) =
    w_c . mean(x), E[y_c^2] = w_c^T E[x x^T] w_c - against the same statistics of the materialised y in float64. x has a mean far
    from zero in some channels (the variance is a difference of two large numbers there)."""
    import ctypes
    from orbit_dataset_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(Cin + P)
    x = torch.randn(P, Cin, generator=g) * (0.5 + torch.rand(Cin, generator=g)) + torch.randn(Cin, generator=g) * 1.5
    w = torch.randn(C, Cin, generator=g) / Cin ** 0.5
    y = x.double() @ w.double().t()
    want_mean, want_var = y.mean(0), y.var(0, unbiased=False)
    eps = 1e-3
    xd, wd = x.to(device), w.to(device)
    mean, invstd = torch.empty(C, device=device), torch.empty(C, device=device)
    _lib.check(lib.orbit_op_bn_stats_from_gram(_lib.dptr(xd), P, Cin, _lib.dptr(wd), C, ctypes.c_float(eps), _lib.dptr(mean),
                                               _lib.dptr(invstd), _lib.stream_handle()), "orbit_op_bn_stats_from_gram")
    got_var = 1.0 / invstd.cpu().double() ** 2 - eps
    assert (mean.cpu().double() - want_mean).abs().max().item() <= 2e-6 * max(1.0, want_mean.abs().max().item())
    assert ((got_var - want_var).abs() / (want_var + eps)).max().item() <= 2e-5
