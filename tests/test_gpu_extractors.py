"""GPU parity of the native feature extractors / set encoder / FiLM generator against the PyTorch-CPU oracle
(same deterministic parameters, same seeded synthetic frames), through orbit_extractor_* / orbit_filmgen_*."""
import pytest
import torch
from torch.func import functional_call

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from oracle import blocks, extractors  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.model.feature_adapters import FilmParameterGenerator  # noqa: E402
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor  # noqa: E402
from orbit_dataset_amd.model.film import get_film_parameter_sizes, get_film_parameters  # noqa: E402
from orbit_dataset_amd.model.set_encoders import SetEncoder  # noqa: E402

FEAT_TOL = 2e-5  # fp32: absolute on O(1) features, relative to the largest feature beyond that


def feat_err(got, want):
    return (got - want).abs().max().item() / max(1.0, want.abs().max().item())


def _frames(n, size, seed=3):
    t = synthetic.make_task(seed, way=4, shots=1, frames_per_shot=-(-n // 4), num_query=1, frame_size=size)
    return t["context_clips"].flatten(end_dim=1)[:n]


def _pair(name, with_film=False):
    ref = extractors.create(name).eval()
    synthetic.init_parameters_(ref)
    fe, film_names = create_feature_extractor(name, True, with_film, False)
    fe.load_state_dict(ref.state_dict())
    return ref, fe.cuda().eval(), film_names  # like ref: running-statistics BatchNorm (nn.Module defaults to train())


@pytest.mark.parametrize("name,size,n", [("resnet18", 84, 12), ("resnet18", 32, 5), ("resnet18", 224, 3),
                                         ("resnet18", 97, 2), ("efficientnet_b0", 224, 4),
                                         ("efficientnet_b0", 192, 6), ("efficientnet_b0", 231, 2)])
def test_extractor_matches_oracle(device, name, size, n):
    ref, fe, _ = _pair(name)
    x = _frames(n, size)
    with torch.no_grad():
        want = ref(x)
    got = fe(x.to(device)).cpu()
    assert got.shape == want.shape == (n, fe.output_size)
    assert torch.isfinite(want).all() and want.abs().max().item() < 50, "oracle features left the calibrated regime"
    err = feat_err(got, want)
    assert err < FEAT_TOL, f"{name}@{size}: max feature err {err} (feature max {want.abs().max().item():.3f})"


@pytest.mark.parametrize("name,size,amp", [("resnet18", 84, 0.2), ("efficientnet_b0", 224, 0.03)])
def test_extractor_film_matches_functional_call(device, name, size, amp):
    """Per-task FiLM: the reference swaps BatchNorm weight/bias by name via functional_call
    (few_shot_recognisers.py:114-115); both the fast path (film=) and the functional_call path must agree."""
    ref, fe, film_names = _pair(name, with_film=True)
    g = torch.Generator().manual_seed(11)
    params = dict(ref.named_parameters())
    film = {}
    for n_ in film_names:
        p = params[n_].detach()
        film[n_] = p * (1 + amp * torch.randn(p.shape, generator=g)) + 0.25 * amp * torch.randn(p.shape, generator=g)
    x = _frames(4, size)
    with torch.no_grad():
        want = functional_call(ref, film, (x,))
        plain = ref(x)
    assert (want - plain).abs().max().item() > 1e-2  # FiLM really changes the features
    assert want.abs().max().item() < 50
    film_dev = {k: v.to(device) for k, v in film.items()}
    got_fc = functional_call(fe, film_dev, (x.to(device),)).cpu()
    slots = [n_ for n_, _ in fe.film_slot_modules()]
    gamma = torch.cat([film_dev[s + ".weight"] for s in slots])
    beta = torch.cat([film_dev[s + ".bias"] for s in slots])
    got_fast = fe(x.to(device), film=(gamma, beta)).cpu()
    assert feat_err(got_fc, want) < FEAT_TOL
    assert torch.equal(got_fc, got_fast)
    assert feat_err(fe(x.to(device)).cpu(), plain) < FEAT_TOL  # and the un-FiLMed path is untouched


@pytest.mark.parametrize("size,n", [(84, 10), (224, 3), (32, 4)])
def test_set_encoder_matches_oracle(device, size, n):
    ref = blocks.SetEncoder().eval()
    synthetic.init_parameters_(ref)
    enc = SetEncoder()
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda().eval()
    x = _frames(n, size)
    with torch.no_grad():
        want = ref(x)
    got = enc(x.to(device))
    assert (got.cpu() - want).abs().max().item() < FEAT_TOL
    z = enc.aggregate(got, "mean").cpu()
    assert z.shape == (1, 64) and (z - ref.aggregate(want)).abs().max().item() < FEAT_TOL


def test_film_generator_matches_oracle(device):
    ref_fe, fe, film_names = _pair("efficientnet_b0", with_film=True)
    sizes = get_film_parameter_sizes(film_names, fe)
    init = get_film_parameters(film_names, fe)
    gen = FilmParameterGenerator(sizes, init, 64, 64, slot_names=[n for n, _ in fe.film_slot_modules()]).cuda()
    synthetic.init_parameters_(gen, prefix="film_generator.")
    ref = blocks.FilmParameterGenerator({k: v for k, v in sizes.items()}, {k: v.cpu() for k, v in init.items()})
    ref.load_state_dict({k: v.cpu() for k, v in gen.state_dict().items()})
    assert ref.film_parameter_names == gen.film_parameter_names and len(gen.film_parameter_names) == 34
    z = torch.randn(1, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = ref(z)
    got = gen(z.to(device))
    assert set(got) == set(want)
    for k in want:
        assert (got[k].cpu() - want[k]).abs().max().item() < 1e-5, k
    assert abs(float(gen.regularization_term()) - float(ref.regularization_term())) < 1e-4 * float(ref.l2_term)
    assert sum(v.numel() for v in got.values()) == 20480


def test_graph_replay_matches_eager(device):
    """ORBIT_GRAPH=1: the captured-and-replayed forward must equal the eager launch sequence bit for bit (run in a
    subprocess because the switch is read once per process)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import orbit_dataset_amd\n"
        "from orbit_dataset_amd import synthetic\n"
        "from orbit_dataset_amd.model.feature_extractors import create_feature_extractor\n"
        "fe,_ = create_feature_extractor('resnet18', True, False, False); synthetic.init_parameters_(fe); fe = fe.cuda().eval()\n"
        "x = torch.randn(6, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()\n"
        "out = torch.empty(6, 512, device='cuda')\n"
        "outs = [fe(x, out=out).clone() for _ in range(4)]  # eager, capture, replay, replay\n"
        "torch.cuda.synchronize(); print(all(torch.equal(outs[0], o) for o in outs), float(outs[0].abs().sum()))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flag in ("0", "1", "2"):
        env = dict(os.environ, ORBIT_GRAPH=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = r.stdout.strip().splitlines()[-1]
    assert res["0"].startswith("True") and res["0"] == res["1"] == res["2"], res


def test_largest_forward_of_the_reference_configuration(device):
    """batch_size 256 clips x clip_length 8 = 2048 frames of 224x224 in ONE forward: the biggest activation has 2.5e9
    elements (> 2^31), so this pins the 64-bit indexing of every kernel; rows are independent, hence bit-identical to the
    same frames pushed through in 8 forwards of 256."""
    fe, _ = create_feature_extractor("efficientnet_b0", True, False, False)
    synthetic.init_parameters_(fe)
    fe = fe.cuda().eval()
    x = torch.randn(2048, 3, 224, 224, device=device, generator=torch.Generator(device=device).manual_seed(1))
    with torch.no_grad():
        big = fe(x)
        small = torch.cat([fe(x[i:i + 256]) for i in range(0, 2048, 256)])
    assert torch.isfinite(big).all() and torch.equal(big, small)


@pytest.mark.parametrize("graph", [1])
def test_extractor_with_bf16x3_pointwise_convs_matches_oracle(device, graph):
    """`conv_bf3` (opt-in, csrc/conv_bf3.hip): EfficientNet-B0's 14x14 / 7x7 pointwise convs on the bf16 matrix cores with three-way
    split operands. Same oracle, same bound as the default path (FEAT_TOL); the features differ from the default path's (another
    summation), and an option change re-captures a replayed launch sequence (the option epoch is part of the graph key)."""
    from orbit_dataset_amd import _lib
    lib = _lib.load()
    ref, fe, _ = _pair("efficientnet_b0")
    x = _frames(4, 224)
    with torch.no_grad():
        want = ref(x)
    xd = x.to(device)
    prev, prev_graph = lib.orbit_get_option(b"conv_bf3"), lib.orbit_get_option(b"graph")
    try:
        lib.orbit_set_option(b"graph", graph)
        lib.orbit_set_option(b"conv_bf3", 0)
        out = torch.empty(4, fe.output_size, device=device)
        base = [fe(xd, out=out).clone() for _ in range(3)][-1]  # eager, capture, replay
        lib.orbit_set_option(b"conv_bf3", 3)  # dense convs (bit 1) and the expand stage of the row-streaming fronts (bit 2)
        got = [fe(xd, out=out).clone() for _ in range(3)]
        lib.orbit_set_option(b"conv_bf3", 0)
        back = fe(xd, out=out).clone()
    finally:
        lib.orbit_set_option(b"conv_bf3", prev)
        lib.orbit_set_option(b"graph", prev_graph)
    assert feat_err(base.cpu(), want) < FEAT_TOL
    assert all(torch.equal(got[0], g_) for g_ in got) and not torch.equal(got[0], base)  # the split kernels did run
    assert feat_err(got[0].cpu(), want) < FEAT_TOL
    assert torch.equal(back, base)


@pytest.mark.parametrize("name,frames,opt", [("efficientnet_b0", 160, 0), ("efficientnet_b0", 160, 3), ("resnet18", 96, 0)])
def test_two_extractors_on_two_streams_repeat_their_solo_results(device, name, frames, opt):
    """Two independent plans, each on its own stream, issued back to back so their kernels share the chip: every forward must
    return the bits of its solo run. With `conv_bf3` this caught a kernel that was correct alone: fragment reads of the next
    k-step overwrote the operand registers of issued-but-queued v_mfma_f32_32x32x16_bf16 instructions when another stream's
    matrix instructions delayed them (features off by up to 1e-2 from run to run; csrc/conv_bf3.hip reads every fragment of a
    K-tile before its first MFMA since)."""
    from orbit_dataset_amd import _lib
    lib = _lib.load()
    fes = []
    for seed in (0, 3):
        fe, _ = create_feature_extractor(name, True, False, False)
        synthetic.init_parameters_(fe, seed=seed)
        fes.append(fe.cuda().eval())
    g = torch.Generator(device=device).manual_seed(7)
    xs = [torch.randn(frames, 3, 224, 224, device=device, generator=g) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    prev, prev_graph = lib.orbit_get_option(b"conv_bf3"), lib.orbit_get_option(b"graph")
    try:
        lib.orbit_set_option(b"conv_bf3", opt)
        lib.orbit_set_option(b"graph", 0)
        with torch.no_grad():
            solo = [fes[i](xs[i]).clone() for i in range(2)]
            torch.cuda.synchronize()
            for rep in range(12):
                outs = []
                for i in range(2):
                    with torch.cuda.stream(streams[i]):
                        outs.append(fes[i](xs[i]))
                torch.cuda.synchronize()
                for i in range(2):
                    assert torch.equal(outs[i], solo[i]), "rep %d plan %d: max diff %g" % (rep, i, (outs[i] - solo[i]).abs().max().item())
    finally:
        lib.orbit_set_option(b"conv_bf3", prev)
        lib.orbit_set_option(b"graph", prev_graph)
