"""Worker of tests/test_gpu_dist.py: one rank of a 2- / 4- / 8-rank job in which ALL ranks share GPU 0 (gloo backend: RCCL refuses
two ranks on one device). Runs the PRODUCT's multi-GPU forms - dist.personalise_support_sharded, dist.predict_query_sharded,
learner.py --mode train with dist.GradientBucket - and writes what it computed to <out>.rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import dist as odist  # noqa: E402
from orbit_dataset_amd import synthetic  # noqa: E402


def _sharded_case(case):
    """(extractor, way, frames per class, query frames, label values) of the support/query-sharded tests"""
    if case == "effnet10":  # BASELINE config 5's head shape: 10-way, D = 1280 -> the [C*D + C] = 12 810-float payload
        return "efficientnet_b0", 10, 4, 13, None
    return "resnet18", 4, 5, 11, (2, 5, 6, 9)


def sharded(out, adapt, case="resnet4"):
    from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser
    rank, world, _ = odist.init_from_env("gloo")
    torch.cuda.set_device(0)
    fe, way, per_class, nq, values = _sharded_case(case)
    model = SingleStepFewShotRecogniser(fe, adapt, "proto", 1, 8, False, 16, 1.0)
    synthetic.init_parameters_(model)
    model._set_device("cuda:0")
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(9, way=way, shots=1, frames_per_shot=per_class, num_query=nq, frame_size=64,
                               label_values=values)
    ctx, lab, tgt = task["context_clips"].cuda(), task["context_labels"].cuda(), task["target_clips"].cuda()
    sh = odist.SupportSharding(rank, world)
    lo, hi = sh.bounds(len(lab))
    # this rank only holds ITS slice of the support clips (plus the full, tiny label vector)
    odist.personalise_support_sharded(model, ctx[lo:hi].clone(), lab, sh)
    logits = odist.predict_query_sharded(model, tgt, sh)
    torch.save({"W": model.classifier.weight.detach().cpu(), "b": model.classifier.bias.detach().cpu(),
                "logits": logits.cpu(), "bounds": (lo, hi)}, "%s.rank%d.pt" % (out, rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def rank_ordered_sum(x, world):
    """((x_0 + x_1) + x_2) + ... in fp32 on the host: the order both P2P kernels promise. With more than two addends a
    different order gives different bits, so equality with this is a real statement about the kernel's summation order."""
    parts = [torch.empty_like(x) for _ in range(world)]
    torch.distributed.all_gather(parts, x)
    acc = parts[0].clone()
    for q in range(1, world):
        acc = acc + parts[q]
    return acc


def train(out, argv):
    from orbit_dataset_amd import learner
    args = learner.build_parser().parse_args(argv + ["--save_model_path", out + ".model.pt"])
    L = learner.Learner(args)
    stats = L.run()
    grads_none = L.gradless_parameters  # parameters whose grad was None at the first optimizer step
    torch.save({"stats": stats, "bucket_bytes": getattr(L.grad_bucket, "nbytes", 0) if L.world > 1 else 0,
                "grads_none": grads_none}, "%s.rank%d.pt" % (out, L.rank))
    if L.world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def p2p(out):
    """One-shot P2P all-reduce between `world` processes that share GPU 0: IPC-mapped inboxes, flags, rank-order sums."""
    import time
    rank, world, _ = odist.init_from_env("gloo")
    torch.cuda.set_device(0)
    ar = odist.P2PAllReduce(rank, world, max_floats=16384)
    res = {"memory_kind": ar.memory_kind}
    g = torch.Generator().manual_seed(100 + rank)
    # prototype payloads (5 x 1280 + 5; 10 x 1280 + 10 = BASELINE config 5), embedding sums, a scalar, a full slot
    for n in (6405, 12810, 65, 1, 16384, 6405):
        x = torch.randn(n, generator=g) * (10.0 ** (rank % 3))  # mixed magnitudes: the summation order shows in the bits
        want = rank_ordered_sum(x, world)
        dev = x.cuda()
        ar(dev)
        torch.cuda.synchronize()
        res.setdefault("got", []).append(dev.cpu())
        res.setdefault("want", []).append(want)
    # the sharded product path through the P2P exchange
    from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser
    model = SingleStepFewShotRecogniser("resnet18", True, "proto", 1, 8, False, 16, 1.0)
    synthetic.init_parameters_(model)
    model._set_device("cuda:0")
    model._send_to_device()
    model.set_test_mode(True)
    task = synthetic.make_task(9, way=4, shots=1, frames_per_shot=5, num_query=11, frame_size=64, label_values=(2, 5, 6, 9))
    ctx, lab, tgt = task["context_clips"].cuda(), task["context_labels"].cuda(), task["target_clips"].cuda()
    sh = odist.SupportSharding(rank, world, p2p=ar)
    lo, hi = sh.bounds(len(lab))
    odist.personalise_support_sharded(model, ctx[lo:hi].clone(), lab, sh)
    res["W"] = model.classifier.weight.detach().cpu()
    res["logits"] = model.predict(tgt).cpu()
    # latency of the 25.6 KB exchange (per-call, host-synchronised: an upper bound)
    dev = torch.randn(6405).cuda()
    for _ in range(5):
        ar(dev)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(50):
        ar(dev)
    torch.cuda.synchronize()
    res["us_per_allreduce"] = 1e6 * (time.perf_counter() - t0) / 50
    res["error"] = ar.error()
    torch.save(res, "%s.rank%d.pt" % (out, rank))
    torch.distributed.barrier()
    ar.close()
    torch.distributed.destroy_process_group()


def p2p_timeout(out):
    """A peer that never shows up: the waiting rank's kernel gives up after ~4 s, its buffer comes back as NaN (not as a
    partial sum) and the error word names the missing rank; P2PAllReduce.raise_on_error turns that into an exception."""
    rank, world, _ = odist.init_from_env("gloo")
    torch.cuda.set_device(0)
    ar = odist.P2PAllReduce(rank, world, max_floats=4096)
    x = torch.full((1000,), float(rank + 1)).cuda()
    ar(x)  # epoch 1: both ranks
    torch.cuda.synchronize()
    res = {"first": x.cpu(), "error_before": ar.error()}
    if rank == 0:
        y = torch.ones(1000).cuda()
        ar(y)  # epoch 2: rank 1 never calls
        torch.cuda.synchronize()
        res["poisoned"] = y.cpu()
        res["error_after"] = ar.error()
        try:
            ar.raise_on_error()
            res["raised"] = False
        except RuntimeError as e:
            res["raised"] = str(e)
    torch.save(res, "%s.rank%d.pt" % (out, rank))
    torch.distributed.barrier()
    ar.close()
    torch.distributed.destroy_process_group()


def p2p_bucket(out):
    """Sharded P2P all-reduce (direct reduce-scatter + all-gather) of gradient-bucket-sized vectors, `world` processes on GPU 0."""
    import time
    rank, world, _ = odist.init_from_env("gloo")
    torch.cuda.set_device(0)
    big = 5_288_548 + 64 * 213  # efficientnet_b0's parameters in 256-byte slots
    ar = odist.P2PAllReduce(rank, world, max_floats=odist.P2PAllReduce.floats_for_bucket(big, world))
    res = {"got": [], "want": [], "memory_kind": ar.memory_kind}
    g = torch.Generator().manual_seed(200 + rank)
    # odd lengths: ragged last shard / last slice, n % world != 0, shards shorter than the 64-block grid
    for n in (big, 32769, 100001, 1_000_003, big, 40000, 32771):
        x = torch.randn(n, generator=g) * (10.0 ** (rank % 3))
        want = rank_ordered_sum(x, world)
        dev = x.cuda()
        ar(dev)
        torch.cuda.synchronize()
        res["got"].append(dev.cpu())
        res["want"].append(want)
    # interleaved with the one-shot form on the same inbox (epochs are shared)
    total = float(world * (world + 1) // 2)
    small = torch.full((6405,), float(rank + 1)).cuda()
    ar(small)
    dev = torch.ones(big).cuda() * (rank + 1)
    ar(dev)
    torch.cuda.synchronize()
    res["mixed_small"], res["mixed_big"], res["mixed_total"] = small.cpu(), dev.cpu(), total
    dev = torch.randn(big).cuda()
    for _ in range(3):
        ar(dev)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        ar(dev)
    torch.cuda.synchronize()
    res["us_per_allreduce"] = 1e6 * (time.perf_counter() - t0) / 20
    res["bytes"] = 4 * big
    res["error"] = ar.error()
    torch.save(res, "%s.rank%d.pt" % (out, rank))
    torch.distributed.barrier()
    ar.close()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "sharded":
        sharded(out, adapt=sys.argv[3] == "1", case=sys.argv[4] if len(sys.argv) > 4 else "resnet4")
    elif mode == "p2p_timeout":
        p2p_timeout(out)
    elif mode == "p2p":
        p2p(out)
    elif mode == "p2p_bucket":
        p2p_bucket(out)
    elif mode == "train":
        train(out, sys.argv[3:])
    else:
        raise SystemExit("unknown mode " + mode)
