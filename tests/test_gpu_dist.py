"""The product's multi-rank forms executed by 2 and 8 processes on hardware (all ranks on GPU 0, gloo backend - the GPU
box has one device and RCCL refuses two ranks on it; the exchange steps are the same torch.distributed calls the nccl
backend serves on a multi-GPU node; the peer-to-peer kernels of csrc/comm.hip run at the node's real world size, 8):
  * dist.personalise_support_sharded + dist.predict_query_sharded: each rank extracts features for ITS slice of the
    support clips, ONE all-reduce of the [C*D + C] prototype payload, bit-identical prototypes on both ranks, equal to
    the single-process personalise()/predict();
  * learner.py --mode train (LITE, tasks dealt round-robin, dist.GradientBucket all-reduce per optimizer step): the
    parameters after the step equal single-process accumulation over the same tasks (reference tasks_per_batch
    semantics, single-step-learner.py:162-166,231), parameters without a gradient stay untouched."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, args, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", ORBIT_DIST_BACKEND="gloo",
                   OMP_NUM_THREADS="8", MKL_NUM_THREADS="8")  # (8 ranks x one thread per CPU of a 256-CPU host thrash)
        procs.append(subprocess.Popen([sys.executable, WORKER] + args, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs


def _single_process_reference(device, fe, adapt, way, per_class, nq, values):
    model = SingleStepFewShotRecogniser(fe, adapt, "proto", 1, 8, False, 16, 1.0)
    synthetic.init_parameters_(model)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(9, way=way, shots=1, frames_per_shot=per_class, num_query=nq, frame_size=64,
                               label_values=values)
    with torch.no_grad():
        model.personalise(task["context_clips"].cuda(), task["context_labels"].cuda())
        want = model.predict(task["target_clips"].cuda()).cpu()
    return model.classifier.weight.detach().cpu(), want


@pytest.mark.parametrize("world,adapt,case", [(2, False, "resnet4"), (2, True, "resnet4"), (8, False, "effnet10")])
def test_support_and_query_sharded_on_ranks(device, world, adapt, case, tmp_path):
    """dist.personalise_support_sharded / predict_query_sharded on 2 and 8 ranks (BASELINE config 5 splits 8 ways):
    ragged support slices (40 clips over 8 ranks), the 10-way D = 1280 prototype payload, prototypes
    bit-identical on all ranks and equal (fp32 rounding of a different summation order) to the single-process run."""
    out = str(tmp_path / "sh")
    _launch(world, ["sharded", out, "1" if adapt else "0", case])
    rs = [torch.load("%s.rank%d.pt" % (out, r)) for r in range(world)]
    fe, way, per_class, nq, values = ("efficientnet_b0", 10, 4, 13, None) if case == "effnet10" else \
        ("resnet18", 4, 5, 11, (2, 5, 6, 9))
    n = way * per_class
    assert [r["bounds"] for r in rs] == [_bounds(n, r, world) for r in range(world)]
    for r in rs[1:]:  # bit-identical prototypes and logits on every rank
        assert torch.equal(rs[0]["W"], r["W"]) and torch.equal(rs[0]["b"], r["b"])
        assert torch.equal(rs[0]["logits"], r["logits"])
    W, want = _single_process_reference(device, fe, adapt, way, per_class, nq, values)
    assert rs[0]["W"].shape == W.shape
    assert (rs[0]["W"] - W).abs().max().item() < 1e-5 * max(1.0, W.abs().max().item())
    assert (rs[0]["logits"] - want).abs().max().item() < 1e-3
    assert torch.equal(rs[0]["logits"].argmax(1), want.argmax(1))


def _bounds(n, rank, world):
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return (lo, lo + q + (1 if rank < r else 0))


def _test_logits(device, recipe, state_dict):
    model = SingleStepFewShotRecogniser(recipe[1], False, "proto", 1, 8, False, 4, 1.0)
    model.load_state_dict(state_dict)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(77, way=3, shots=1, frames_per_shot=6, num_query=20, frame_size=64)
    with torch.no_grad():
        model.personalise(task["context_clips"].cuda(), task["context_labels"].cuda())
        return model.predict(task["target_clips"].cuda()).cpu()


TRAIN = ["--mode", "train", "--with_lite", "--num_lite_samples", "4", "--frame_size", "64", "--way", "3", "--shots", "1",
         "--frames_per_shot", "4", "--num_query_videos", "2", "--frames_per_video", "5", "--batch_size", "8",
         "--num_train_tasks", "4", "--tasks_per_batch", "4", "--optimizer", "sgd", "--learning_rate", "0.05",
         "--weight_decay", "0.1"]


@pytest.mark.parametrize("recipe", [["--feature_extractor", "resnet18", "--learn_extractor"],
                                    ["--feature_extractor", "resnet18", "--adapt_features", "--classifier", "versa"],
                                    ["--feature_extractor", "efficientnet_b0", "--learn_extractor"]],
                         ids=["protonet_resnet18", "cnaps_versa_resnet18", "protonet_efficientnet_b0"])
def test_task_parallel_training_step_equals_single_process(device, recipe, tmp_path):
    one, two = str(tmp_path / "w1"), str(tmp_path / "w2")
    _launch(1, ["train", one] + TRAIN + recipe)
    _launch(2, ["train", two] + TRAIN + recipe)
    a, b = torch.load(one + ".model.pt"), torch.load(two + ".model.pt")
    init = SingleStepFewShotRecogniser(recipe[1], "--adapt_features" in recipe,
                                       recipe[recipe.index("--classifier") + 1] if "--classifier" in recipe else "proto",
                                       1, 8, "--learn_extractor" in recipe, 4, 1.0)
    synthetic.init_parameters_(init, film_strength=0.02 if recipe[1] == "efficientnet_b0" else 0.1)
    init_sd = init.state_dict()
    moved, stat_worst = 0, 0.0
    for k in a:
        if k.endswith("num_batches_tracked"):
            assert int(a[k]) == int(b[k]), k  # every rank counts the whole window's forwards
            continue
        if k.endswith(("running_mean", "running_var")):
            # train-mode statistics: the ranks' windows are combined as sequential updates would have been
            # (dist.RunningStatSync) - what is left is the recency order of the tasks inside the window, a fraction of the
            # distance the window moved the statistic (round 2 averaged the ranks' values: unbounded)
            win = (a[k].float() - init_sd[k].float()).abs().max().item()
            d = (a[k].float() - b[k].float()).abs().max().item()
            stat_worst = max(stat_worst, d / max(win, 1e-6))
            assert d <= 1e-5 + 0.5 * win, "%s: 1 vs 2 ranks differ by %g, the window moved it by %g" % (k, d, win)
            continue
        d = (a[k].float() - b[k].float()).abs().max().item()
        step = (a[k].float() - init_sd[k].float()).abs().max().item()
        assert d <= 2e-6 + 2e-3 * step, "%s differs between 1 and 2 ranks by %g (step size %g)" % (k, d, step)
        moved += step > 0
    assert moved > 10
    if "--learn_extractor" in recipe:
        print("running statistics, 1 vs 2 ranks: worst difference = %.3f of the window's movement" % stat_worst)
        # and what that does to a model: test-mode logits of the two trained models on a held-out task
        la, lb = _test_logits(device, recipe, a), _test_logits(device, recipe, b)
        if recipe[1] == "resnet18":  # (one SGD step at lr 0.05 sends the efficientnet_b0 recipe's test-mode features to inf
            scale = la.abs().max().item()  # in the single-process run too: nothing to compare there)
            assert torch.isfinite(la).all() and (la - lb).abs().max().item() <= 0.02 * scale, (
                (la - lb).abs().max().item(), scale)
            assert (la.argmax(1) == lb.argmax(1)).float().mean().item() >= 0.95
    info1, info2 = torch.load(one + ".rank0.pt"), torch.load(two + ".rank0.pt")
    assert info2["bucket_bytes"] > 0
    # parameters that get no gradient in the single-process run get none in the 2-rank run either
    assert info1["grads_none"] == info2["grads_none"]
    for k in info2["grads_none"]:
        assert torch.equal(b[k], init_sd[k].to(b[k].dtype)), k


# ---- BASELINE config 5's split: 10-way tasks, tasks_per_batch 16, on 2 / 8 ranks, against one rank AND the oracle --------
C5 = dict(way=10, frames_per_shot=2, num_query_videos=2, frames_per_video=5, frame_size=64, batch_size=8, num_lite=4,
          num_train_tasks=20, tasks_per_batch=16, lr=0.002, weight_decay=0.1)
C5_ARGS = ["--mode", "train", "--with_lite", "--num_lite_samples", str(C5["num_lite"]), "--frame_size", str(C5["frame_size"]),
           "--way", str(C5["way"]), "--shots", "1", "--frames_per_shot", str(C5["frames_per_shot"]), "--num_query_videos",
           str(C5["num_query_videos"]), "--frames_per_video", str(C5["frames_per_video"]), "--batch_size", str(C5["batch_size"]),
           "--num_train_tasks", str(C5["num_train_tasks"]), "--tasks_per_batch", str(C5["tasks_per_batch"]), "--optimizer", "sgd",
           "--learning_rate", str(C5["lr"]), "--weight_decay", str(C5["weight_decay"]),
           "--feature_extractor", "efficientnet_b0", "--learn_extractor"]
C5_RECIPE = ["--feature_extractor", "efficientnet_b0", "--learn_extractor"]


def _oracle_config5_training():
    """The same 20 tasks / 2 optimizer steps through oracle/training.py (PyTorch-CPU autograd restatement of
    single-step-learner.py:212-243, pinned by goldens G6 / G8 / G9): same initial weights, tasks, LITE permutations
    (np.random seeded per task as learner.py does), loss scaling, SGD. Returns the trained extractor's state_dict."""
    import numpy as np
    from oracle.recogniser import OracleRecogniser
    from oracle.training import LiteTrainer
    seed = synthetic.DEFAULT_SEED
    init = SingleStepFewShotRecogniser("efficientnet_b0", False, "proto", 1, C5["batch_size"], True, C5["num_lite"], 1.0)
    synthetic.init_parameters_(init, seed=seed, film_strength=0.02)
    ref = OracleRecogniser("efficientnet_b0", False, "proto", 1, C5["batch_size"], num_lite_samples=C5["num_lite"])
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v.clone() for k, v in init.state_dict().items()
                            if k.startswith("feature_extractor.")})
    trainer = LiteTrainer(ref, True, C5["tasks_per_batch"])
    opt = torch.optim.SGD(list(ref.fe.parameters()), lr=C5["lr"], momentum=0.0, weight_decay=C5["weight_decay"])
    opt.zero_grad()
    total = C5["num_train_tasks"]
    for step in range(total):
        t = synthetic.make_task(10_000 + step, C5["way"], 1, C5["frames_per_shot"], C5["num_query_videos"] * C5["frames_per_video"],
                                C5["frame_size"], clip_length=1, seed=seed)
        np.random.seed((seed + 7919 * (step + 1)) % (2 ** 32))
        trainer.train_task_with_lite(t["context_clips"], t["context_labels"], t["target_clips"], t["target_labels"])
        if (step + 1) % C5["tasks_per_batch"] == 0 or step == total - 1:
            opt.step()
            opt.zero_grad()
    return {k: v.detach().clone() for k, v in ref.fe.state_dict().items()}


def _oracle_test_logits(fe_state):
    from oracle.recogniser import OracleRecogniser
    ref = OracleRecogniser("efficientnet_b0", False, "proto", 1, 8)
    ref.fe.load_state_dict(fe_state)
    task = synthetic.make_task(77, way=3, shots=1, frames_per_shot=6, num_query=20, frame_size=64)
    ref.personalise(task["context_clips"], task["context_labels"])
    return ref.predict(task["target_clips"])


@pytest.fixture(scope="module")
def config5_single(tmp_path_factory):
    from concurrent.futures import ThreadPoolExecutor
    out = str(tmp_path_factory.mktemp("c5") / "w1")
    with ThreadPoolExecutor(1) as ex:  # the one-rank GPU run (a subprocess) beside the CPU oracle's replay
        run = ex.submit(_launch, 1, ["train", out] + C5_ARGS)
        oracle = _oracle_config5_training()
        run.result()
    return torch.load(out + ".model.pt"), oracle


@pytest.mark.parametrize("world", [2, 8])
def test_task_parallel_training_config5_split(device, world, config5_single, tmp_path):
    """BASELINE config 5's partitioning (reference single-step-learner.py:162-166,231: an optimizer step every tasks_per_batch
    = 16 tasks, 10-way tasks) on 2 and 8 ranks: 20 tasks = one full window (16: two tasks per rank at world 8) and a ragged
    one (4 tasks: at world 8 ranks 4-7 run NOTHING in it, 20 % 8 != 0) - gradients all-reduced per step, BatchNorm running
    statistics combined over ranks that ran 0, 1 or 2 forwards. efficientnet_b0 at a learning rate at which the trained
    model stays finite (round 3's lr 0.05 sent its test-mode features to inf), so the comparison reaches the MODEL: parameters
    N ranks vs one, test-mode logits N ranks vs one, and both against the same training replayed through the CPU oracle."""
    single, oracle_fe = config5_single
    out = str(tmp_path / ("w%d" % world))
    _launch(world, ["train", out] + C5_ARGS, timeout=900)
    multi = torch.load(out + ".model.pt")
    init = SingleStepFewShotRecogniser("efficientnet_b0", False, "proto", 1, C5["batch_size"], True, C5["num_lite"], 1.0)
    synthetic.init_parameters_(init, film_strength=0.02)
    init_sd = init.state_dict()
    moved, stat_vs_oracle = 0, 0.0
    for k in single:
        a, b = single[k].float(), multi[k].float()
        if k.endswith("num_batches_tracked"):
            assert int(single[k]) == int(multi[k]), k
            continue
        if k.endswith(("running_mean", "running_var")):
            # the single-process product against the oracle's replay: every running statistic, directly (a fraction of how
            # far the two windows moved it)
            o = oracle_fe[k[len("feature_extractor."):]].float()
            win_o = (a - init_sd[k].float()).abs().max().item()
            stat_vs_oracle = max(stat_vs_oracle, (a - o).abs().max().item() / max(win_o, 1e-6))
            assert (a - o).abs().max().item() <= 1e-5 + 5e-2 * win_o, "%s: product vs oracle training" % k
            # Combined as sequential updates would have been (dist.RunningStatSync), up to the ORDER in which the recency
            # weights fall on the tasks: one process weights the window's last task most, N ranks weight their last tasks
            # alike. With ~4 train-mode forwards per task and a 4-task last window that is a visible share of how far the
            # window moved a near-zero statistic (measured 0.55 for bn1.running_mean at world 2), so the bound here is the
            # window's movement itself; what it does to the MODEL is bounded through the test-mode logits below.
            win = (a - init_sd[k].float()).abs().max().item()
            assert (a - b).abs().max().item() <= 1e-5 + 1.0 * win, k
            continue
        step = (a - init_sd[k].float()).abs().max().item()
        assert (a - b).abs().max().item() <= 2e-6 + 2e-3 * step, "%s: 1 vs %d ranks" % (k, world)
        moved += step > 0
        # ... and the single-process product against the oracle's replay of the same two optimizer steps
        o = oracle_fe[k[len("feature_extractor."):]].float()
        # (worst element measured: 0.54 % of the layer's largest update, on a 7x7-stage filter whose train-mode BatchNorm sees
        # 2x2 maps at this frame size; the per-gradient parity of one step is pinned at full size by G8 / G9)
        assert (a - o).abs().max().item() <= 2e-6 + 2e-2 * step, "%s: product vs oracle training" % k
    assert moved > 50
    # The model. Two optimizer windows from a synthetic initialisation leave the running statistics far from the batch
    # statistics (momentum 0.1), so test-mode logits are huge (~1e6) and magnify any difference in the statistics through 81
    # eval-mode BatchNorms: the PARAMETERS' effect is compared with the statistics held equal (the N-rank parameters under the
    # one-rank statistics), the statistics' own recency-order difference is bounded above and its effect printed.
    la = _test_logits(device, C5_RECIPE, single)
    lb_raw = _test_logits(device, C5_RECIPE, multi)
    same_stats = {k: (single[k] if k.endswith(("running_mean", "running_var", "num_batches_tracked")) else v)
                  for k, v in multi.items()}
    lb = _test_logits(device, C5_RECIPE, same_stats)
    lo = _oracle_test_logits(oracle_fe)
    scale = lo.abs().max().item()
    assert torch.isfinite(la).all() and torch.isfinite(lb).all() and torch.isfinite(lb_raw).all() and torch.isfinite(lo).all()
    print("world %d: test-mode logit scale %.3g; 1 vs N ranks |dlogit| %.3g with equal statistics, %.3g with each run's own; "
          "product vs oracle %.3g (worst running statistic %.2e of its movement off the oracle's)"
          % (world, scale, (la - lb).abs().max().item(), (la - lb_raw).abs().max().item(), (la - lo).abs().max().item(),
             stat_vs_oracle))
    assert (la - lb).abs().max().item() <= 0.02 * scale, ((la - lb).abs().max().item(), scale)
    assert (la - lb_raw).abs().max().item() <= 0.25 * scale  # the recency order of the window's statistics (see above)
    assert (la - lo).abs().max().item() <= 0.02 * scale, ((la - lo).abs().max().item(), scale)
    assert (la.argmax(1) == lb.argmax(1)).float().mean().item() >= 0.95
    assert (la.argmax(1) == lo.argmax(1)).float().mean().item() >= 0.95


@pytest.mark.parametrize("world", [2, 8])
def test_one_shot_p2p_allreduce(device, world, tmp_path):
    """csrc/comm.hip orbit_p2p_*: `world` processes map each other's inbox through HIP IPC (all on GPU 0 here; over xGMI on
    a multi-GPU node), push + flag + rank-order sum. The results are BIT-IDENTICAL on every rank and equal, bit for bit, to
    ((x0 + x1) + x2) + ... computed on the host - with 4 and 8 addends of mixed magnitude that pins the summation order
    (two addends commute). Payloads: the 5-way and the 10-way D = 1280 prototype vectors, 65 embedding sums, a
    scalar, a full slot. The inbox is uncached / fine-grained device memory; successive epochs do not
    interfere; dist.personalise_support_sharded through the exchange gives the single-process prototypes."""
    out = str(tmp_path / "p2p")
    _launch(world, ["p2p", out], timeout=400)
    rs = [torch.load("%s.rank%d.pt" % (out, r)) for r in range(world)]
    for r in rs:
        assert r["error"] == 0
        assert r["memory_kind"] in (1, 2), "the inbox must be uncached or fine-grained device memory"
        for got, got0, want in zip(r["got"], rs[0]["got"], rs[0]["want"]):
            assert torch.equal(got, got0)   # identical bits on every rank
            assert torch.equal(got, want)   # and exactly the rank-ordered sum
        assert torch.equal(r["W"], rs[0]["W"]) and torch.equal(r["logits"], rs[0]["logits"])
    _, want = _single_process_reference(device, "resnet18", True, 4, 5, 11, (2, 5, 6, 9))
    assert (rs[0]["logits"] - want).abs().max().item() < 1e-3 and torch.equal(rs[0]["logits"].argmax(1), want.argmax(1))
    print("one-shot P2P all-reduce of 25.6 KB between %d processes on one GPU: %s us per call (host-synchronised)"
          % (world, " / ".join("%.1f" % r["us_per_allreduce"] for r in rs)))


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_p2p_allreduce_of_gradient_buckets(device, world, tmp_path):
    """csrc/comm.hip orbit_p2p_allreduce_sum_sharded (direct reduce-scatter + all-gather): vectors of the
    gradient bucket's size (21 MB) and ragged lengths (n % world != 0, last shard / last slice short, shards shorter than
    the 64-block grid); bit-identical on all ranks and equal to the rank-ordered host sum, repeatable across epochs and
    interleaved with the one-shot form on the same inbox. World 8 = 8 x 64 spinning blocks co-resident on one GPU."""
    out = str(tmp_path / "p2pb")
    _launch(world, ["p2p_bucket", out], timeout=600)
    rs = [torch.load("%s.rank%d.pt" % (out, r)) for r in range(world)]
    for r in rs:
        assert r["error"] == 0 and r["memory_kind"] in (1, 2)
        for got, want in zip(r["got"], rs[0]["want"]):
            assert torch.equal(got, want)
        assert torch.equal(r["mixed_small"], torch.full_like(r["mixed_small"], r["mixed_total"]))
        assert torch.equal(r["mixed_big"], torch.full_like(r["mixed_big"], r["mixed_total"]))
    print("sharded P2P all-reduce of %.1f MB between %d processes on one GPU: %s us per call"
          % (rs[0]["bytes"] / 1e6, world, " / ".join("%.0f" % r["us_per_allreduce"] for r in rs)))


def test_p2p_timeout_poisons_the_buffer_and_raises(device, tmp_path):
    """a peer that never arrives must not yield a silently wrong sum. The waiting kernel gives up after ~4 s,
    returns NaN instead of the partial sum, and the host sees the error without a device-wide hipMemcpy."""
    out = str(tmp_path / "p2pt")
    _launch(2, ["p2p_timeout", out], timeout=300)
    r0, r1 = torch.load(out + ".rank0.pt"), torch.load(out + ".rank1.pt")
    assert torch.equal(r0["first"], torch.full((1000,), 3.0)) and torch.equal(r1["first"], r0["first"])
    assert r0["error_before"] == 0 and r1["error_before"] == 0
    assert bool(torch.isnan(r0["poisoned"]).all())
    assert r0["error_after"] == 2  # 1 + the rank whose flag never came
    assert r0["raised"] and "rank 1" in r0["raised"]


def test_rccl_communicator_behind_the_c_abi(device):
    """csrc/comm.hip orbit_comm_* (RCCL): a one-rank communicator on this box's GPU - unique id, init, all-reduce, destroy.
    (World > 1 needs one GPU per rank: bench.py --gpus N runs the same check over all ranks and reports `rccl_ranks`.)"""
    import ctypes
    from orbit_dataset_amd import _lib
    code = ("import ctypes, torch, orbit_dataset_amd\n"
            "from orbit_dataset_amd import _lib\n"
            "lib = _lib.load()\n"
            "uid = ctypes.create_string_buffer(128)\n"
            "_lib.check(lib.orbit_comm_unique_id(uid), 'id')\n"
            "_lib.check(lib.orbit_comm_init(0, 1, uid), 'init')\n"
            "assert lib.orbit_comm_world() == 1 and lib.orbit_comm_rank() == 0\n"
            "x = torch.arange(6405, dtype=torch.float32, device='cuda')\n"
            "_lib.check(lib.orbit_allreduce_sum(_lib.dptr(x), x.numel(), _lib.stream_handle()), 'allreduce')\n"
            "torch.cuda.synchronize()\n"
            "assert torch.equal(x.cpu(), torch.arange(6405, dtype=torch.float32))\n"
            "assert lib.orbit_comm_init(0, 1, uid) != 0  # already initialised\n"
            "lib.orbit_comm_destroy()\n"
            "assert lib.orbit_comm_world() == 0\n"
            "print('rccl ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_training_step_with_p2p_gradient_bucket(device, tmp_path):
    """learner.py --p2p_gradients: the flat gradient bucket goes through the sharded P2P all-reduce; the trained model
    equals the 2-rank run that uses the backend's all-reduce (two addends: bit for bit)."""
    recipe = ["--feature_extractor", "resnet18", "--learn_extractor"]
    ring, p2p = str(tmp_path / "ring"), str(tmp_path / "p2p")
    _launch(2, ["train", ring] + TRAIN + recipe)
    _launch(2, ["train", p2p] + TRAIN + recipe + ["--p2p_gradients"])
    a, b = torch.load(ring + ".model.pt"), torch.load(p2p + ".model.pt")
    for k in a:
        assert torch.equal(a[k], b[k]), k
