#!/usr/bin/env python
"""Throughput of the episodic few-shot hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU. Launched by torch.distributed.run it reads
                                                          RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; launched plainly it
                                                          re-executes itself under torch.distributed.run with N ranks. A
                                                          line is only ever printed with n_gpus == --gpus)

One step = one synthetic ORBIT-shaped task through SingleStepFewShotRecogniser.personalise() + predict() with
the support/query frames already resident in HBM (5-way, 5 shots x 8 frames = 200 support frames, 200 query
frames). Metric (BASELINE.json): query frames/sec per task = M / (t_personalise + t_predict); `value` is the
whole-job rate over all ranks (tasks are independent units: rank r runs its own tasks, no data-path collective;
"scaling": "weak"). Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel = the fp32-MFMA
implicit-GEMM convolution, per-launch HIP events on its stream) and `cpu_baseline` (the PyTorch-CPU oracle of
the same path timed on this box's host cores, N=1 only, bounded sample).

`--mode lite_train` (resnet18 workloads) times the LITE meta-training step instead: one step = one task through
Learner.train_task_with_lite (per query batch: personalise_with_lite -> predict_a_batch -> scaled CE -> backward)
plus the optimizer step; with N > 1 every step all-reduces (RCCL) one flat gradient bucket. Same metric / unit /
JSON contract; the roofline aggregates the forward + data-gradient (conv_igemm) and weight-gradient (conv_wgrad)
kernels.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import _lib, synthetic  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402

WORKLOADS = {
    # name: (extractor, adapt_features, frame size)   — BASELINE.json configs[1..3]
    "resnet18_84": ("resnet18", False, 84),
    "efficientnet_b0_224": ("efficientnet_b0", False, 224),
    "resnet18_224": ("resnet18", False, 224),
    "cnaps_resnet18_224": ("resnet18", True, 224),
    # the README's other two single-step recipes (heads of SURVEY §8f rank 2), inference only
    "cnaps_versa_resnet18_224": ("resnet18", True, 224),
    "simple_cnaps_resnet18_224": ("resnet18", True, 224),
}
HEADS = {"cnaps_versa_resnet18_224": "versa", "simple_cnaps_resnet18_224": "mahalanobis"}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
WAY, SHOTS, FRAMES_PER_SHOT, NUM_QUERY = 5, 5, 8, 200


NUM_LITE = 16


def build_model(workload, device, batch_size=256, train=False):
    fe_name, adapt, _ = WORKLOADS[workload]
    # meta-training recipes of the reference README: ProtoNets learn the extractor; CNAPs keep it frozen and learn the
    # set encoder + FiLM generator
    learn_extractor = bool(train and not adapt)
    model = SingleStepFewShotRecogniser(fe_name, adapt, HEADS.get(workload, "proto"), 1, batch_size, learn_extractor,
                                        NUM_LITE, 1.0)
    synthetic.init_parameters_(model)  # (also re-takes the FiLM generator's snapshot of the extractor's BatchNorm)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(not train)
    # the query pass of predict() overlaps the support pass of personalise() on a second HIP stream (inputs are resident
    # in HBM before the timed region, so they are ready whenever predict() is called); ORBIT_BENCH_OVERLAP=0 disables it
    # ORBIT_BENCH_OVERLAP=2: pipelined form (head on the query stream, no join: consecutive tasks overlap out of phase)
    # Round 6: "1" leaves the recogniser in its DEFAULT mode (overlap_query = "auto": the query pass overlaps whenever the clips
    # are known to be ready) - bench marks its resident query clips ready right after it has produced them
    # (data.utils.mark_ready), which is all a caller of personalise() / predict() has to do to get `value`
    mode = os.environ.get("ORBIT_BENCH_OVERLAP", "1")
    model.overlap_query = False if (train or mode == "0") else (2 if mode == "2" else "auto")
    # LITE: the query batch's taped pass starts beside the cache pass as well (same readiness requirement as overlap_query:
    # the clips are resident); ORBIT_LITE_OVERLAP=0 / ORBIT_LITE_QUERY_OVERLAP=0 restore the serial order
    model.lite_query_overlap = bool(train and model.lite_overlap and os.environ.get("ORBIT_LITE_QUERY_OVERLAP", "1") != "0")
    return model


class LiteTrainStep:
    """Learner.train_task_with_lite (reference single-step-learner.py:212-243) for ONE task per call, and the optimizer
    step of the reference's outer loop (:162-166) after every `tasks_per_rank` calls: an optimizer step covers
    tasks_per_batch = tasks_per_rank x world tasks (BASELINE config 5: 8 x 8 = 64), every task's loss carries
    1/tasks_per_batch (:231), and with N > 1 ranks the gradients are summed by ONE all-reduce of the persistent flat
    bucket (dist.GradientBucket) right before the step."""

    def __init__(self, model, world, batch_size, tasks_per_rank=1):
        from orbit_dataset_amd import dist as odist
        from orbit_dataset_amd.learner import init_optimizer
        self.model, self.world, self.batch_size = model, world, batch_size
        self.tasks_per_rank, self.calls = int(tasks_per_rank), 0
        self.window_loss, self.step_losses = None, []  # device scalars: this rank's summed loss per optimizer step
        from argparse import Namespace
        # torch's fused multi-tensor Adam (same update formula, one kernel per group instead of ~10 foreach launches over
        # every parameter tensor: the LITE step is host-bound, measured 44.8 -> 42.2 ms on efficientnet_b0, 10.2 -> 9.5 ms on
        # resnet18@84); init_optimizer then invalidates the native plans from a step hook because the fused kernel does not
        # bump parameter versions. ORBIT_BENCH_FUSED_ADAM=0 selects the plain (foreach) optimizer of the reference.
        opt_args = Namespace(fused_optimizer=os.environ.get("ORBIT_BENCH_FUSED_ADAM", "1") == "1")
        self.optimizer = init_optimizer(model, 5e-6, "adam", opt_args, 1.0)
        self.bucket = None
        if world > 1:
            p2p = None
            if os.environ.get("ORBIT_BENCH_P2P_GRADIENTS", "0") == "1":  # direct RS + AG instead of the backend's ring
                cap = sum(-(-q.numel() // 64) * 64 for q in model.parameters() if q.requires_grad)
                p2p = odist.P2PAllReduce(int(os.environ.get("RANK", 0)), world, odist.P2PAllReduce.floats_for_bucket(cap, world))
            self.bucket = odist.GradientBucket(model.parameters(), p2p=p2p)
        import numpy as np
        np.random.seed(1991)
        self._np = np

    def __call__(self, model, task):
        from orbit_dataset_amd.optim import cross_entropy
        if "task_index" in task:  # LITE's permutation is seeded per TASK, so a run does not depend on how tasks are dealt to ranks
            self._np.random.seed((1991 + 7919 * (task["task_index"] + 1)) % (2 ** 32))
        ctx, lab, tgt, tlab = task["context_clips"], task["context_labels"], task["target_clips"], task["target_labels"]
        model._clear_caches()
        out = []
        tasks_per_batch = self.tasks_per_rank * self.world
        with torch.enable_grad():
            for lo in range(0, len(tgt), self.batch_size):
                model.personalise_with_lite(ctx, lab)
                logits = model.predict_a_batch(tgt[lo:lo + self.batch_size])
                loss = len(lab) / (NUM_LITE * tasks_per_batch) * cross_entropy(logits, tlab[lo:lo + self.batch_size])
                loss = loss + 0.001 * model.film_generator.regularization_term()
                loss.backward()
                out.append(logits.detach())
                self.window_loss = loss.detach() if self.window_loss is None else self.window_loss + loss.detach()
                model._reset()
        self.calls += 1
        if self.calls % self.tasks_per_rank == 0:
            self.step_losses.append(self.window_loss)
            self.window_loss = None
            if self.bucket is not None:
                self.bucket.sync()
            self.optimizer.step()
            if self.bucket is not None:
                self.bucket.zero_()
            else:
                self.optimizer.zero_grad()
        return torch.cat(out)


def trained_checkpoint(workload):
    """The meta-trained checkpoint of a workload (tools/meta_train.py: LITE meta-training through the native kernels on
    synthetic tasks of the "blobs" family; fp16-rounded values, so the HIP path and the CPU oracle load identical numbers)
    or None. ORBIT_BENCH_WEIGHTS=synthetic keeps the deterministic random initialisation of rounds 1-2."""
    if os.environ.get("ORBIT_BENCH_WEIGHTS", "trained") != "trained" or HEADS.get(workload):
        return None
    fe_name, adapt, size = WORKLOADS[workload]
    path = os.path.join(ROOT, "orbit-dataset_amd", "assets", "meta_trained_%s_%d.npz" % (fe_name, size))
    return path if (not adapt and os.path.exists(path)) else None


def load_trained_checkpoint(model, path):
    import numpy as np
    sd = {k: torch.from_numpy(v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in np.load(path).items()}
    model.load_state_dict(sd)


def parity_gate(model, fe_name, device):
    """Cheap gate that runs on EVERY rank before the clock starts: the extractor on the committed frames of
    tests/golden/G12_extractors_hf.npz must reproduce the features Hugging Face transformers computed for them (an
    implementation independent of this repository, tests/hf_pin.py) to 2e-5. The full logits-vs-oracle gate follows at
    N = 1 (cpu_baseline leg); either failing means no timing line is printed."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "G12_extractors_hf.npz")
    if not os.path.exists(path):
        return {"fixture": None}
    g = dict(np.load(path))
    case = {"efficientnet_b0": "efficientnet_b0_224", "resnet18": "resnet18_224"}[fe_name]
    x = torch.from_numpy(g["x231"].astype("float32")[:, :, :224, :224].copy()).to(device)
    fe = model.feature_extractor
    was_training = fe.training
    fe.eval()
    with torch.no_grad():
        got = fe(x).cpu()
    fe.train(was_training)
    err = float((got - torch.from_numpy(g[case + "_feats"])).abs().max())
    if not err < 2e-5:
        print("PARITY GATE FAILED (extractor vs transformers fixture %s): max |d feature| %g" % (case, err), file=sys.stderr)
        raise SystemExit(3)
    return {"fixture": "tests/golden/G12_extractors_hf.npz:" + case, "max_abs_dfeature_vs_transformers": err, "tol": 2e-5}


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo; the literal is the fallback)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "query frames/sec per task (224x224, 5-way ProtoNet) + frame accuracy vs ref"


def kernel_sources_sha16():
    """Fingerprint of the HIP sources: a PMC traffic figure measured in another process is only quoted for these."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "orbit-dataset_amd", "csrc")
    for name in sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h"))):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode()), h.update(f.read())
    return h.hexdigest()[:16]


def run_task(model, task):
    with torch.no_grad():  # as every test-time caller of the reference does (single-step-learner.py:311)
        model.personalise(task["context_clips"], task["context_labels"])
        logits = model.predict(task["target_clips"])
    model._reset()
    return logits


def head_roofline(device, n_tasks=64, M=200, D=1280, C=5, reps=40):
    """HBM roofline of the distance kernel (orbit_proto_predict) on the 64-task batched launch SURVEY §8(d) names:
    algorithmic bytes 4*(M*D + C*D + C + M*C) per task = 67.4 MB per launch. Eight distinct query sets (540 MB) are
    cycled so the 256 MB Infinity Cache cannot serve the stream. Timed with one HIP-event pair PER LAUNCH on the launch
    stream (median): a ctypes launch costs the host more than this kernel runs, so timing a loop of launches as a whole
    measures the host. The same kernel on 256 / 1024-task launches (270 MB / 1.08 GB) shows what it streams at once the
    fixed launch cost is amortised."""
    lib = _lib.load()

    def measure(nt, sets):
        g = torch.Generator(device=device).manual_seed(7)
        qs = [torch.rand(nt, M, D, device=device, generator=g) for _ in range(sets)]
        W = torch.rand(nt, C, D, device=device, generator=g)
        b = torch.rand(nt, C, device=device, generator=g)
        out = torch.empty(nt, M, C, device=device)

        def run(i):
            _lib.check(lib.orbit_proto_predict(_lib.dptr(qs[i % sets]), _lib.dptr(W), _lib.dptr(b), nt, M, 1, D, C, 1.0, 0,
                                               _lib.dptr(out), None, _lib.stream_handle()), "orbit_proto_predict")
        for i in range(sets):
            run(i)
        evs = []
        for i in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(i)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(1e3 * a.elapsed_time(b_) for a, b_ in evs)
        us = ts[len(ts) // 2]
        nbytes = 4.0 * (M * D + C * D + C + M * C) * nt
        return us, nbytes, nbytes / (us * 1e-6) / 1e9

    us, nbytes, gbs = measure(n_tasks, 8)
    larger = {}
    for nt, sets in ((256, 4), (1024, 2)):
        u2, b2, g2 = measure(nt, sets)
        larger["%d_tasks" % nt] = {"avg_launch_us": u2, "bytes_per_launch": b2, "achieved_GBps": g2, "frac": g2 / 8000.0}
    traffic, source = None, "none"
    for tname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_head_traffic.json")),
                        reverse=True):  # the latest round's PMC passes of this kernel
        with open(os.path.join(ROOT, "profiles", tname)) as f:
            tj = json.load(f)
        if tj.get("kernel_sources_sha16") == kernel_sources_sha16():
            traffic = tj.get("traffic_bytes_per_launch")
            source = "file profiles/%s (rocprofv3 --pmc passes)" % tname
        else:
            source = "refused: profiles/%s was measured on other kernel sources" % tname
        break
    return {"kernel": "orbit::proto_predict_stream_kernel<8 waves, 2 rows> (64 tasks x 200 queries x 1280, euclidean, 5-way)",
            "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "avg_launch_us": us,
            "bytes_per_launch": nbytes, "traffic": traffic, "traffic_source": source,
            "timing": "median of %d per-launch HIP-event pairs" % reps, "larger_launches": larger}


def metric_variants(model, tasks, device, steps):
    """The two side measurements SURVEY.md §8(d) asks for next to `value` (never reported as `value`):
    predict-only frames/s (one personalise(), then predict() of the 200 query frames repeatedly) and the H2D-inclusive
    rate (support and query clips start in pinned host memory and are uploaded per mini-batch inside the timed region,
    602 KB per 224x224 frame, as the reference's loops do)."""
    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    task = tasks[0]
    with torch.no_grad():
        model.personalise(task["context_clips"], task["context_labels"])
        pred = lambda i: model.predict(tasks[i % len(tasks)]["target_clips"])
        timed(pred, 3)
        t_pred = timed(pred, steps)
    model._reset()
    host = [{k: (v.cpu().pin_memory() if isinstance(v, torch.Tensor) and k.endswith("clips") else v) for k, v in t.items()}
            for t in tasks[:2]]
    h2d = lambda i: run_task(model, host[i % len(host)])
    timed(h2d, 2)
    t_h2d = timed(h2d, steps)
    # the same with the clips held as 8-bit frames on the host (random bytes: throughput only) and normalised on the GPU
    shape = lambda t, k: t[k].shape
    host8 = [{k: (torch.randint(0, 256, shape(t, k), dtype=torch.uint8).pin_memory() if k.endswith("clips") else v)
              for k, v in t.items()} for t in tasks[:2]]
    h2d8 = lambda i: run_task(model, host8[i % len(host8)])
    timed(h2d8, 2)
    t_h2d8 = timed(h2d8, steps)
    # ... and through the input pipeline (data/pipeline.TaskPrefetcher): a staging thread uploads the 8-bit frames of task
    # i+1 on a copy stream and normalises them there while the extractor runs task i (pinned ring of 3 slots)
    from orbit_dataset_amd.data.pipeline import TaskPrefetcher

    def prefetched(n):
        pf = TaskPrefetcher((host8[i % len(host8)] for i in range(n)), device, depth=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in pf:
            run_task(model, t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pf.close()
        return dt
    prefetched(3)
    t_pf = prefetched(steps)

    def prefetched_f32(n):  # the reference's fp32 clips (602 KB per frame) through the same pipeline: upload under compute
        pf = TaskPrefetcher((host[i % len(host)] for i in range(n)), device, depth=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in pf:
            run_task(model, t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pf.close()
        return dt
    prefetched_f32(3)
    t_pf32 = prefetched_f32(steps)
    return {"predict_only_query_frames_per_s": NUM_QUERY * steps / t_pred,
            "h2d_inclusive_query_frames_per_s": NUM_QUERY * steps / t_h2d,
            "h2d_inclusive_fp32_prefetched_query_frames_per_s": NUM_QUERY * steps / t_pf32,
            "h2d_inclusive_uint8_query_frames_per_s": NUM_QUERY * steps / t_pf,
            "h2d_inclusive_uint8_unpipelined_query_frames_per_s": NUM_QUERY * steps / t_h2d8,
            "note": "predict-only: predict() after one personalise(); h2d-inclusive: clips uploaded from pinned host memory inside the "
                    "timed region (fp32 inline / fp32 or 8-bit through data/pipeline.TaskPrefetcher / 8-bit unpipelined): DESIGN.md section 7"}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, model, train=False, way=WAY, template="noise"):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload - the median of 3 tasks
    at 224x224, of 5 at 84x84 (ORBIT_BENCH_CPU_TASKS overrides the count) - on the host cores."""
    from oracle.recogniser import OracleRecogniser
    fe_name, adapt, size = WORKLOADS[workload]
    ref = OracleRecogniser(fe_name, adapt, HEADS.get(workload, "proto"), 1, 256, num_lite_samples=NUM_LITE)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    if HEADS.get(workload) == "versa":
        for name in ("weight_processor", "bias_processor"):
            getattr(ref, name).load_state_dict({k[len("classifier.%s." % name):]: v for k, v in sd.items()
                                                if k.startswith("classifier.%s." % name)})
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v for k, v in sd.items() if k.startswith("feature_extractor.")})
    if adapt:
        ref.set_encoder.load_state_dict({k[len("set_encoder."):]: v for k, v in sd.items() if k.startswith("set_encoder.")})
        ref.build_film_generator().load_state_dict(
            {k[len("film_generator."):]: v for k, v in sd.items() if k.startswith("film_generator.")})
    n_tasks = int(os.environ.get("ORBIT_BENCH_CPU_TASKS", "3" if size >= 224 else "5"))
    if train:
        n_tasks = 1  # (the parity leg replays this one training step on the GPU from the same weights)
    cpu_tasks = [synthetic.make_task(i, way, 1, WAY * SHOTS * FRAMES_PER_SHOT // way, NUM_QUERY, size, template=template)
                 for i in range(max(1, n_tasks))]
    task = cpu_tasks[0]
    # Thread count: BASELINE.md asks for os.cpu_count(); PyTorch-CPU gets SLOWER past the point where the small
    # convolutions stop scaling, so every candidate up to and including os.cpu_count() is timed on a probe with the batch
    # shape of the real task (one 64-frame extractor batch + one 64-frame query batch; the round-1 probe used 32 frames,
    # which under-fed the wide settings) and the fastest is used; `cores` reports the count actually used.
    ncpu = os.cpu_count() or 1
    warm = synthetic.make_task(1, WAY, 1, 12, 64, size)
    probe = {}
    candidates = [int(c) for c in os.environ.get("ORBIT_BENCH_CPU_THREADS", "8,16,32,64,128,%d" % ncpu).split(",")]
    for nt in sorted({min(c, ncpu) for c in candidates}):
        torch.set_num_threads(nt)
        if not probe:
            ref.personalise(warm["context_clips"][:8], warm["context_labels"][:8])  # first-call set-up, untimed
        t0 = time.perf_counter()
        ref.personalise(warm["context_clips"], warm["context_labels"])
        ref.predict(warm["target_clips"])
        probe[nt] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    if train:
        import numpy as np
        from oracle.training import LiteTrainer
        trainer = LiteTrainer(ref, not adapt, 1)
        np.random.seed(1991)
        t0 = time.perf_counter()
        with torch.enable_grad():
            (logits, _), = trainer.train_task_with_lite(task["context_clips"], task["context_labels"],
                                                        task["target_clips"], task["target_labels"])
        dt = time.perf_counter() - t0
        what = "1 LITE training task (200 support + 200 query frames, H=%d, forward + backward)" % NUM_LITE
        per_task = [dt]
    else:
        per_task, logits = [], None
        for t in cpu_tasks:
            t0 = time.perf_counter()
            ref.personalise(t["context_clips"], t["context_labels"])
            lg = ref.predict(t["target_clips"])
            per_task.append(time.perf_counter() - t0)
            if logits is None:
                logits = lg  # the parity leg compares the GPU path on the FIRST task
        dt = sorted(per_task)[len(per_task) // 2]
        what = "median of %d tasks (200 support + 200 query frames each)" % len(per_task)
    return {"value": NUM_QUERY / dt, "unit": "query frames/s", "cores": cores, "kind": "port",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model_name(),
            "per_task_seconds": [round(x, 2) for x in per_task],
            "sample": "%s, %dx%d, PyTorch-CPU oracle on %s, %d threads (fastest of %s on a 60+64-frame probe), "
                      "%.1f s per task, %.0f s of CPU work" % (what, size, size, cpu_model_name(), cores,
                                                              "/".join(str(k) for k in sorted(probe)), dt, sum(per_task)),
            "thread_probe_frames_per_s": {str(k): round(124 / v, 1) for k, v in sorted(probe.items())},
            }, task, logits


def collect_prof(lib):
    """Rows of the per-launch HIP-event records (orbit_prof_*): one per kernel name, with summed duration, algorithmic FLOP /
    bytes, the launch-by-launch roofline floor and the SiLU evaluations."""
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
    _lib.check(lib.orbit_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), "orbit_prof_collect")
    rows = []
    for i in range(lib.orbit_prof_num_variants()):
        name = ctypes.create_string_buffer(48)
        ln, vms, vfl, vby = ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        fm, fs, sl = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.orbit_prof_variant(i, name, ctypes.byref(ln), ctypes.byref(vms), ctypes.byref(vfl), ctypes.byref(vby))
        lib.orbit_prof_variant_floor(i, ctypes.byref(fm), ctypes.byref(fs), ctypes.byref(sl))
        if ln.value and vms.value > 0:
            rows.append({"name": name.value.decode(), "launches": ln.value, "ms": vms.value, "flops": vfl.value,
                         "bytes": vby.value, "floor_ms": fm.value, "floor_simd_ms": fs.value, "silu": sl.value})
    return rows


def conv_rows_summary(rows):
    """The dense-convolution rows (the `roofline` kernel family) and the other profiled kernels, as rounds 1-5 reported them."""
    variants, other, total_bytes, n_main, conv_ms, conv_fl = [], [], 0.0, 0, 0.0, 0.0
    for r in rows:
        sec = r["ms"] * 1e-3
        tf, gbs = r["flops"] / sec / 1e12, r["bytes"] / sec / 1e9
        if not r["name"].startswith("conv"):
            other.append({"kernel": r["name"], "launches": r["launches"], "avg_us": round(1e3 * r["ms"] / r["launches"], 2),
                          "tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1)})
            continue
        conv_ms += r["ms"]
        conv_fl += r["flops"]
        # the split-K reduce pass of a conv is part of that conv's time and bytes, not a launch of its own
        n_main += 0 if r["name"].startswith("conv_splitk_reduce") else r["launches"]
        total_bytes += r["bytes"]
        variants.append({"kernel": r["name"], "launches": r["launches"], "avg_us": round(1e3 * r["ms"] / r["launches"], 2),
                         "tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1),
                         # which roof the algorithmic work of this variant sits under (157.3 TFLOP/s vs 8 TB/s)
                         "binding_roof": "mfma" if tf / PEAK_FP32_MFMA_TFLOPS >= gbs / 8000.0 else "hbm",
                         "frac_of_binding_roof": round(max(tf / PEAK_FP32_MFMA_TFLOPS, gbs / 8000.0), 3)})
    return variants, other, total_bytes, n_main, conv_ms, conv_fl


FAMILIES = (("dense_conv", ("conv_igemm", "conv_splitk_reduce", "conv_pw_rgemm", "conv_bf3")),
            ("conv_wgrad", ("conv_wgrad",)), ("fused_front", ("mbconv_rows",)), ("stem", ("stem",)),
            ("depthwise", ("dwconv",)), ("se_gate", ("se_gate",)), ("head", ("head_",)))


def family_table(rows, tasks_profiled, ms_per_step):
    """EVERY kernel family of the task against its own roofline floor. Per family and task: launches, time,
    algorithmic FLOP and bytes, floor = sum over its launches of max(bytes / 6.3 TB/s, FLOP / 157.3 TFLOP/s), and the same
    floor with the SiLU evaluations' VALU time added to the matrix time (`floor_simd`: a gfx950 SIMD issues either an MFMA or
    VALU instructions, so a kernel that evaluates SiLU on what its MFMAs produce cannot hide one under the other)."""
    fam = {}
    for r in rows:
        key = next((k for k, prefixes in FAMILIES if r["name"].startswith(prefixes)), "other")
        f = fam.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "floor_ms": 0.0, "floor_simd_ms": 0.0,
                                 "silu": 0.0, "kernels": []})
        for k in ("launches", "ms", "flops", "bytes", "floor_ms", "floor_simd_ms", "silu"):
            f[k] += r[k]
        f["kernels"].append(r["name"])
    n = float(max(tasks_profiled, 1))
    out, tot_us, tot_floor, tot_simd = [], 0.0, 0.0, 0.0
    for key, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        us, fl_us, fs_us = 1e3 * f["ms"] / n, 1e3 * f["floor_ms"] / n, 1e3 * f["floor_simd_ms"] / n
        tot_us, tot_floor, tot_simd = tot_us + us, tot_floor + fl_us, tot_simd + fs_us
        out.append({"family": key, "launches_per_task": round(f["launches"] / n, 1), "us_per_task": round(us, 1),
                    "avg_us": round(1e3 * f["ms"] / f["launches"], 2), "gflop_per_task": round(f["flops"] / n / 1e9, 3),
                    "mb_per_task": round(f["bytes"] / n / 1e6, 2), "silu_mevals_per_task": round(f["silu"] / n / 1e6, 2),
                    "floor_us_per_task": round(fl_us, 1), "x_floor": round(us / fl_us, 2) if fl_us > 0 else None,
                    "floor_simd_us_per_task": round(fs_us, 1), "x_floor_simd": round(us / fs_us, 2) if fs_us > 0 else None})
    return {"families": out,
            "task_kernel_ms_serial": tot_us / 1e3,  # sum of the per-launch durations of one task with every kernel running alone
            "task_floor_ms": tot_floor / 1e3, "whole_task_frac_of_floor": tot_floor / 1e3 / ms_per_step,
            "task_floor_simd_ms": tot_simd / 1e3, "whole_task_frac_of_floor_simd": tot_simd / 1e3 / ms_per_step,
            "floor_definition": "sum over launches of max(bytes / 6.3e12, FLOP / 157.3e12); floor_simd: SiLU evaluations / 5.93e12 per s "
                                "added to the matrix time (DESIGN.md section 7); frac = floor / TIMED step"}


def lite_train_block(args, device, lib, tasks, ckpt, steps=10, warmup=8):
    """the LITE meta-training step (reference single-step-learner.py:212-243) in the DEFAULT run's line:
    >= 10 optimizer steps of the headline extractor timed after the inference legs - forward passes (cache pass, H-subset,
    query batch), backward, fused Adam, one task per step - then the same steps with per-launch events for the dense-conv
    family's fraction of the fp32-MFMA peak. `value` is untouched by it."""
    from orbit_dataset_amd.data.utils import unpack_task
    model = build_model(args.workload, device, args.batch_size, train=True)
    if ckpt:
        load_trained_checkpoint(model, ckpt)
    step = LiteTrainStep(model, 1, args.batch_size, 1)
    host_labels = [t["context_labels"].cpu() for t in tasks]
    import gc
    gc.collect()
    gc.freeze()  # (the new model's objects leave the cyclic collector's working set, as in main())

    def stream(n):  # a new device label tensor per task, its label set taken from the host copy (as the training loop does)
        return [dict(tasks[i % len(tasks)],
                     context_labels=unpack_task({"context_labels": host_labels[i % len(tasks)], "target_labels": None,
                                                 "context_clips": None, "target_clips": None}, device)[2]) for i in range(n)]
    for t in stream(warmup):  # (first sight of a call runs eagerly, the second captures its graph, later ones replay: with
        #                        four resident tasks every call has been seen twice after eight steps)
        step(model, t)
    torch.cuda.synchronize()
    step.step_losses = []
    g0 = model.feature_extractor.train_graph_stats()
    todo = stream(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in todo:
        step(model, t)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    g1 = model.feature_extractor.train_graph_stats()
    losses = [float(x) for x in torch.stack(step.step_losses).cpu()] if step.step_losses else []
    # per-launch events with the three forward passes serial (kernels run one at a time; graphs are bypassed while recording)
    ov = (model.lite_overlap, model.lite_query_overlap)
    model.lite_overlap = model.lite_query_overlap = False
    prof_steps = max(2, steps // 2)
    todo = stream(prof_steps)
    lib.orbit_prof_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in todo:
        step(model, t)
    torch.cuda.synchronize()
    prof_elapsed = time.perf_counter() - t0
    lib.orbit_prof_enable(0)
    model.lite_overlap, model.lite_query_overlap = ov
    rows = collect_prof(lib)
    variants, _other, total_bytes, n_main, conv_ms, conv_fl = conv_rows_summary(rows)
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    traffic, traffic_source = pmc_traffic(args.workload + ":lite_train")
    fams = family_table(rows, prof_steps, 1e3 * elapsed / steps)
    return {"ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
            "query_frames_per_s": NUM_QUERY * steps / elapsed,
            "step": "one task through train_task_with_lite (200 + 200 frames, H = %d: cache, H-subset and taped query pass, backward, "
                    "fused Adam), one optimizer step per task" % NUM_LITE,
            "host_enqueue_ms_per_step": 1e3 * issued / steps,
            "train_loss_per_step": losses,
            # calls of the native training entry points INSIDE the timed steps: replayed from captured HIP graphs / run eagerly
            "train_graph_calls_replayed_eager": [g1[0] - g0[0], g1[1] - g0[1]],
            "lite_subset_beside_cache_pass": bool(ov[0]), "lite_query_pass_beside_cache_pass": bool(ov[1]),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": total_bytes / max(n_main, 1), "launches": n_main,
                         "avg_launch_us": 1e3 * conv_ms / max(n_main, 1),
                         "kernel": "dense-convolution family of the step: orbit::conv_igemm_kernel (forward + data gradient), "
                                   "orbit::pw_rgemm_kernel, orbit::conv_wgrad_kernel / conv_wgrad_thin_kernel",
                         "kernel_time_share": conv_ms / (1e3 * prof_elapsed),
                         "measured": "per-launch HIP events over %d further steps, the three forward passes serial, graphs "
                                     "bypassed (%.1f ms/step)" % (prof_steps, 1e3 * prof_elapsed / prof_steps),
                         "variants_in": "python bench.py --mode lite_train (same kernels, per-variant rows)"},
            # (the step's kernels that carry event records: the MFMA families, depthwise, gates, head - not its elementwise passes)
            "instrumented_families": [{k: f[k] for k in ("family", "launches_per_task", "us_per_task", "floor_us_per_task", "x_floor")}
                                      for f in fams["families"] if f["family"] != "other"]}


def pmc_traffic(workload_tag):
    """HBM traffic per launch of a workload's dominant kernel family from the latest committed PMC passes
    (tools/collect_profiles.sh -> profiles/rNN_*traffic.json), refused when the kernel sources changed since."""
    traffic, source = None, "none (no PMC pass recorded for this workload)"
    for tname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
        with open(os.path.join(ROOT, "profiles", tname)) as f:
            tj = json.load(f)
        if tj.get("workload") != workload_tag:
            continue
        if tj.get("kernel_sources_sha16") == kernel_sources_sha16():
            traffic = tj.get("traffic_bytes_per_launch")
            source = "file profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on these " \
                     "kernel sources, %s)" % (tname, tj.get("kernel_sources_sha16"))
        else:
            source = "refused: profiles/%s was measured on other kernel sources (%s, now %s)" % (
                tname, tj.get("kernel_sources_sha16"), kernel_sources_sha16())
        break
    return traffic, source


def full_size_gate(workload, way, device, cores, template, model=None):
    """one FULL-SIZE task (200 support + 200 query frames of 224x224) of a BASELINE config this run does not
    time, HIP path against the CPU oracle: max |d logit| and argmax identity go into the line (and gate it)."""
    from oracle.recogniser import OracleRecogniser
    fe_name, adapt, size = WORKLOADS[workload]
    if model is None:
        model = build_model(workload, device, 256, train=False)
        ckpt = trained_checkpoint(workload)
        if ckpt:
            load_trained_checkpoint(model, ckpt)
    ref = OracleRecogniser(fe_name, adapt, HEADS.get(workload, "proto"), 1, 256, num_lite_samples=NUM_LITE)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v for k, v in sd.items() if k.startswith("feature_extractor.")})
    if adapt:
        ref.set_encoder.load_state_dict({k[len("set_encoder."):]: v for k, v in sd.items() if k.startswith("set_encoder.")})
        ref.build_film_generator().load_state_dict(
            {k[len("film_generator."):]: v for k, v in sd.items() if k.startswith("film_generator.")})
    task = synthetic.make_task(0, way, 1, WAY * SHOTS * FRAMES_PER_SHOT // way, NUM_QUERY, size, template=template)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    ref.personalise(task["context_clips"], task["context_labels"])
    want = ref.predict(task["target_clips"])
    cpu_s = time.perf_counter() - t0
    got = run_task(model, {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}).cpu()
    return {"workload": workload, "way": way, "frames": "200 support + 200 query, %dx%d" % (size, size),
            "max_abs_dlogit_vs_oracle": float((got - want).abs().max().item()),
            "argmax_identical": bool(torch.equal(got.argmax(1), want.argmax(1))),
            "frame_accuracy_gpu": float((got.argmax(1) == task["target_labels"]).float().mean()),
            "frame_accuracy_oracle": float((want.argmax(1) == task["target_labels"]).float().mean()),
            "oracle_seconds": round(cpu_s, 2), "oracle_threads": cores}


def comm_selfcheck(rank, world, backend, dist, device):
    """what the first real multi-GPU run needs in order to be diagnosed from its JSON alone. One-time, outside
    every timed region: the peer-to-peer inbox path (hipIpcGetMemHandle / hipIpcOpenMemHandle over xGMI, csrc/comm.hip) is
    created, its memory kind read, and one all-reduce of the prototype payload pushed through it; errors are reported, not
    raised - the task-parallel data path has no collective and does not depend on it."""
    info = {"rank": rank, "device": torch.cuda.get_device_name(device), "backend": backend,
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    lib = _lib.load()

    def all_ok(flag, what):
        """every rank learns whether the step worked EVERYWHERE before anyone enters the next collective (a rank that failed
        to map a peer's inbox must not leave the others waiting at a barrier)"""
        got = [None] * world
        dist.all_gather_object(got, (bool(flag), "" if flag else _lib.last_error()[:200]))
        bad = ["rank %d: %s" % (r, msg) for r, (ok, msg) in enumerate(got) if not ok]
        if bad:
            info[what] = "failed: " + "; ".join(bad)
        return not bad

    h = ctypes.c_void_p()
    stage = "p2p_create"
    if all_ok(lib.orbit_p2p_create(rank, world, 16384, ctypes.byref(h)) == 0, stage):
        kind = int(lib.orbit_p2p_memory_kind(h))
        info["p2p_memory_kind"] = {1: "uncached device memory", 2: "fine-grained device memory",
                                   3: "coarse-grained (ORBIT_P2P_ALLOW_COARSE)"}.get(kind, str(kind))
        mine = ctypes.create_string_buffer(64)
        stage = "p2p_ipc_export"
        if all_ok(lib.orbit_p2p_export(h, mine) == 0, stage):
            handles = [None] * world
            dist.all_gather_object(handles, bytes(mine.raw))
            table = ctypes.create_string_buffer(b"".join(handles), 64 * world)
            stage = "p2p_ipc_open"
            if all_ok(lib.orbit_p2p_connect(h, table) == 0, stage):
                info["p2p_ipc_open"] = "ok (hipIpcOpenMemHandle of %d peer inboxes)" % (world - 1)
                dist.barrier()  # every inbox is mapped everywhere before the first push
                ones = torch.ones(6405, device=device)  # prototype sums + counts of a 5-way, D = 1280 task
                rc = lib.orbit_p2p_allreduce_sum(h, _lib.dptr(ones), ones.numel(), _lib.stream_handle())
                torch.cuda.synchronize()  # (the kernel's flag wait is a bounded spin: a missing peer poisons, it does not hang)
                lo, hi = float(ones.min().item()), float(ones.max().item())
                info["p2p_allreduce_of_ones"] = [lo, hi]
                info["p2p_allreduce_ok"] = bool(rc == 0 and lo == hi == float(world))
                info["p2p_error_word"] = int(lib.orbit_p2p_error(h))
                dist.barrier()  # nobody unmaps while a peer may still push
    if h:
        lib.orbit_p2p_destroy(h)
    gathered = [None] * world
    dist.all_gather_object(gathered, info)
    return gathered


def rccl_check(lib, rank, world, backend, dist, device):
    """The C-ABI's own RCCL communicator (csrc/comm.hip orbit_comm_init / orbit_allreduce_sum - the carrier a non-PyTorch
    host uses for the prototype / gradient exchange) all-reduces a ones vector over all ranks: the result must be `world`
    in every element. Returns that sum (None when ranks share a GPU under the gloo self-test: RCCL refuses that)."""
    if backend != "nccl":
        return None
    # RCCL prints a version banner on STDOUT when it initialises; this process owes its caller exactly one JSON line there
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return _rccl_check(lib, rank, world, dist, device)
    finally:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # the banner sits in C stdio's buffer: flush it while fd 1 still points at stderr
        os.dup2(saved, 1)
        os.close(saved)


def _rccl_check(lib, rank, world, dist, device):
    uid = ctypes.create_string_buffer(128)
    if rank == 0:
        _lib.check(lib.orbit_comm_unique_id(uid), "orbit_comm_unique_id")
    if world > 1:
        box = [bytes(uid.raw)]
        dist.broadcast_object_list(box, src=0)
        uid = ctypes.create_string_buffer(box[0], 128)
    _lib.check(lib.orbit_comm_init(rank, world, uid), "orbit_comm_init")
    ones = torch.ones(6405, device=device)  # the prototype payload of a 5-way, D = 1280 task
    _lib.check(lib.orbit_allreduce_sum(_lib.dptr(ones), ones.numel(), _lib.stream_handle()), "orbit_allreduce_sum")
    torch.cuda.synchronize()
    lo, hi = float(ones.min().item()), float(ones.max().item())
    lib.orbit_comm_destroy()
    if lo != hi or lo != float(world):
        raise SystemExit("bench.py: orbit_allreduce_sum over %d ranks returned [%g, %g], expected %d" % (world, lo, hi, world))
    return int(lo)


def per_rank_report(rank, world, dist, device, elapsed, issued, steps, train_step):
    """What a SCALE record needs to be read: every rank's own step time and host enqueue time, and - for
    the training form - the gradient bucket's size and the time of ONE all-reduce of it, alone on the chip, through the
    backend (RCCL ring under `nccl`) and through the direct reduce-scatter + all-gather over the P2P inboxes
    (csrc/comm.hip). Collective: every rank calls it; the list is gathered on all ranks."""
    info = {"rank": rank, "ms_per_step": 1e3 * elapsed / steps, "host_enqueue_ms_per_step": 1e3 * issued / steps}
    bucket = getattr(train_step, "bucket", None)
    if bucket is not None and getattr(bucket, "flat", None) is not None:
        from orbit_dataset_amd import dist as odist
        n = bucket.flat.numel()
        info["gradient_bucket_bytes"] = 4 * n
        scratch = torch.zeros_like(bucket.flat)

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e6 * (time.perf_counter() - t0) / reps
        info["allreduce_us_backend"] = timed(lambda: dist.all_reduce(scratch, op=dist.ReduceOp.SUM))
        info["allreduce_algbw_GBps_backend"] = 4e-3 * n / info["allreduce_us_backend"]
        if os.environ.get("ORBIT_BENCH_P2P_PROBE", "1") != "0":
            try:
                p2p = odist.P2PAllReduce(rank, world, odist.P2PAllReduce.floats_for_bucket(n, world))
                info["allreduce_us_p2p"] = timed(lambda: p2p(scratch))
                info["allreduce_algbw_GBps_p2p"] = 4e-3 * n / info["allreduce_us_p2p"]
                p2p.raise_on_error()
                p2p.close()
            except Exception as e:  # (IPC mapping unavailable: report, do not fail the line)
                info["allreduce_p2p_error"] = repr(e)[:200]
    gathered = [None] * world
    dist.all_gather_object(gathered, info)
    return gathered


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)  # fresh boxes with a slow host need ~10 tasks to reach steady state
    ap.add_argument("--workload", default="efficientnet_b0_224", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="inference", choices=["inference", "lite_train"])
    ap.add_argument("--way", type=int, default=5, help="classes per task (BASELINE config 5 is 10-way: 20 support frames "
                                                       "per class instead of 40, same 200 + 200 frames)")
    ap.add_argument("--tasks-per-rank", type=int, default=1,
                    help="lite_train: tasks accumulated per rank and optimizer step (reference --tasks_per_batch = this x "
                         "--gpus; BASELINE config 5 = 8 x 8 GPUs = 64 tasks/step). One bench step = one optimizer step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-tasks", type=int, default=4, help="tasks resident in HBM, cycled through")
    ap.add_argument("--batch-size", type=int, default=256, help="clips per extractor call (reference --batch_size)")
    args = ap.parse_args()

    backend = os.environ.get("ORBIT_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (this form used to run ONE rank
        # and print an N = 1 line). Same command line the driver uses for N > 1.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:  # never a line whose n_gpus differs from what was asked for
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; no line printed" % (args.gpus, world))
    _lib.require_gpu()
    # one rank per GPU; ORBIT_BENCH_BACKEND=gloo lets several ranks share a GPU (self-test of the N>1 path on a
    # one-GPU box — RCCL refuses two ranks on one device); the line then says so (`ranks_share_gpus`)
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible (one rank per GPU over RCCL); no line printed"
                         % (world, torch.cuda.device_count()))
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fe_name, adapt, size = WORKLOADS[args.workload]
    train = args.mode == "lite_train"
    model = build_model(args.workload, device, args.batch_size, train=train)
    gate = parity_gate(model, fe_name, device)  # every rank, before anything is timed (on the synthetic checkpoint the
    #                                             transformers fixture was recorded for)
    ckpt = trained_checkpoint(args.workload)
    template = "blobs" if ckpt else "noise"
    if ckpt:  # the accuracy half of the metric needs weights that carry class signal: the meta-trained checkpoint
        load_trained_checkpoint(model, ckpt)
    run_step = LiteTrainStep(model, world, args.batch_size, args.tasks_per_rank) if train else run_task
    per_step = args.tasks_per_rank if train else 1  # run_step calls (= tasks) per bench step on this rank
    # each rank owns its own tasks (task index = rank + world * i): weak scaling, independent units
    way = args.way
    if (WAY * SHOTS * FRAMES_PER_SHOT) % way:
        raise SystemExit("--way must divide %d support frames" % (WAY * SHOTS * FRAMES_PER_SHOT))
    frames_per_class = WAY * SHOTS * FRAMES_PER_SHOT // way  # keep 200 support frames per task
    tasks = [dict(synthetic.make_task_on_device(rank + world * i, way, 1, frames_per_class, NUM_QUERY, size, 1, device,
                                                template=template), task_index=rank + world * i)
             for i in range(max(1, args.distinct_tasks))]
    # The query clips are resident and complete as of HERE: say so (data.utils.mark_ready records an event on this stream). With
    # that the recogniser's default mode runs the query pass of predict() on its second stream beside the support pass - no
    # opt-in flag on the model (rounds 2-5 set overlap_query = True); ORBIT_BENCH_MARK_READY=0 shows what an unmarked caller gets
    if not train and os.environ.get("ORBIT_BENCH_MARK_READY", "1") != "0":
        from orbit_dataset_amd.data.utils import mark_ready
        for t in tasks:
            mark_ready(t["target_clips"])
    # Every step of every loop below runs on a task whose LABEL TENSOR the head has never seen (a clone made before the clock
    # starts): the per-task label-set resolution - what the reference's configure pays as torch.unique + .item() per task,
    # model/classifier_heads.py:96-100,246-248 - is inside the timed region of `value`. (Round 4 resolved the label sets of
    # the resident tasks before the clock and memoised them; that variant is now reported as `value_memoised_labels`.)
    fresh_labels = os.environ.get("ORBIT_BENCH_FRESH_LABELS", "1") != "0"  # (0: A/B runs of the memoised form only)
    lib = _lib.load()

    host_labels = [t["context_labels"].cpu() for t in tasks] if train else None

    def stream_of_tasks(n):
        if not fresh_labels:
            return [tasks[i % len(tasks)] for i in range(n)]
        if train:
            # the training loop's tasks come from a DataLoader on the host (reference data/queues.py:44-53) and are moved to the
            # device by unpack_task (data/utils.py:30-47): a new device tensor per task, its label set taken from the host copy
            from orbit_dataset_amd.data.utils import unpack_task
            return [dict(tasks[i % len(tasks)],
                         context_labels=unpack_task({"context_labels": host_labels[i % len(tasks)], "target_labels": None,
                                                     "context_clips": None, "target_clips": None}, device)[2]) for i in range(n)]
        return [dict(tasks[i % len(tasks)], context_labels=tasks[i % len(tasks)]["context_labels"].clone()) for i in range(n)]
    # long-lived objects (torch, the model, the resident tasks) leave the cyclic collector's working set: a full
    # collection otherwise walks ~1e6 objects every few steps (measured: 19.7 -> 12.3 ms per LITE step at 84x84)
    import gc
    gc.collect()
    gc.freeze()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def loop(steps):
        todo = stream_of_tasks(steps * per_step)
        barrier()
        from orbit_dataset_amd.model.classifier_heads import PendingLabelSet
        waited0 = PendingLabelSet.wait_seconds
        t0 = time.perf_counter()
        outs = []
        for i in range(steps * per_step):
            outs.append(run_step(model, todo[i]))  # logits stay on the device; scored after the clock stops
        issued = time.perf_counter() - t0  # host time to enqueue everything (diagnostic: host- vs device-bound)
        loop.label_wait = PendingLabelSet.wait_seconds - waited0  # ... of which blocked on a task's label-set count
        barrier()
        elapsed = time.perf_counter() - t0
        # frame accuracy (utils/eval_metrics.py:27-36) outside the timed region: five tiny torch launches per step, and
        # their first call in a fresh process loads code objects (that alone cost 2 ms/step in a first run)
        correct = torch.zeros(2, device=device)
        for i, logits in enumerate(outs):
            correct[0] += (logits.argmax(1) == tasks[i % len(tasks)]["target_labels"]).sum()
            correct[1] += logits.shape[0]
        return elapsed, correct, issued

    # Settling (untimed, before the W warm-up steps): on a fresh box the container image is paged in lazily, and the FIRST
    # process that runs this path stays 25-40 % slower on the host side for its whole life unless the code pages it
    # needs have been touched (measured: first process 17.8 k, every later one 22.3 k query frames/s on the same box).
    # Two chunks of 10 steps at most (capped at 20; the second only if it is still getting faster).
    settling = 0
    if os.environ.get("ORBIT_BENCH_SETTLE", "1") != "0":
        prev = None
        for _ in range(2):
            todo = stream_of_tasks(10 * per_step)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(10 * per_step):
                run_step(model, todo[i])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            settling += 10
            if os.environ.get("ORBIT_BENCH_TRACE"):
                print("settle chunk: %.2f ms/step" % (1e2 * dt), file=sys.stderr)
            done = prev is not None and dt >= 0.97 * prev
            if dist is not None:  # every rank runs the same number of chunks (the training step holds a collective)
                flag = torch.tensor([1.0 if done else 0.0], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                done = bool(flag.item() > 0.5)
            if done:
                break
            prev = dt
    todo = stream_of_tasks(args.warmup * per_step)
    for i in range(args.warmup * per_step):
        run_step(model, todo[i])
    if train:
        run_step.step_losses = []
    elapsed, correct, issued = loop(args.steps)  # the timed region behind `value`
    label_wait = getattr(loop, "label_wait", 0.0)
    train_losses = None
    if train:  # the loss of every optimizer step of the timed region, summed over the ranks (each task's loss already carries
        #        1 / tasks_per_batch, single-step-learner.py:231): N ranks x T tasks must reproduce 1 rank x N T tasks
        train_losses = torch.stack(run_step.step_losses[:args.steps]).double() if run_step.step_losses else torch.zeros(0)
    value_memoised = None
    if not train:  # round 4's form: label sets of the resident tasks resolved and memoised before the clock
        fresh_labels = False
        for t in tasks:
            model.classifier.unique_labels(t["context_labels"], device)
        loop(2)
        m_elapsed, _, _ = loop(args.steps)
        value_memoised = NUM_QUERY * args.steps * per_step / m_elapsed
        fresh_labels = os.environ.get("ORBIT_BENCH_FRESH_LABELS", "1") != "0"
    if train and getattr(run_step.bucket, "p2p", None) is not None:
        run_step.bucket.p2p.raise_on_error()  # (the loop ended on a barrier + synchronize: every exchange has completed)

    def per_task_events(steps):
        """per-task HIP-event time (events on the caller's stream before personalise() and after predict();
        the query stream joins it before the head kernel), median over the tasks of a repeat of the timed steps."""
        todo = stream_of_tasks(steps * per_step)
        barrier()
        evs = []
        for i in range(steps * per_step):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_step(model, todo[i])
            e1.record()
            evs.append((e0, e1))
        barrier()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        return ms[len(ms) // 2]

    timed_mode = getattr(model, "overlap_query", False)  # "auto" (default) | False | 2
    if timed_mode == 2:
        model.overlap_query = True  # per-task latency: the joined form (both passes of ONE task, head on the caller's stream)
    median_task_ms = per_task_events(args.steps)
    value_overlap_off = value_overlap_joined = None
    if bool(timed_mode):
        model.overlap_query = False
        loop(2)
        off_elapsed, _, _ = loop(args.steps)
        value_overlap_off = NUM_QUERY * args.steps * per_step * world / off_elapsed
        if timed_mode == 2:  # and the round-2 form: query pass on the second stream, joined before the head
            model.overlap_query = True
            loop(2)
            j_elapsed, _, _ = loop(args.steps)
            value_overlap_joined = NUM_QUERY * args.steps * per_step * world / j_elapsed
        model.overlap_query = timed_mode
    # roofline leg: the SAME K steps again with one HIP-event pair recorded per conv_igemm launch on its stream
    # (kept out of the timed region above: recording ~80 events per task costs host time and serialises the queue)
    overlap, lite_overlap = getattr(model, "overlap_query", False), getattr(model, "lite_overlap", False)
    lite_query_overlap = getattr(model, "lite_query_overlap", False)
    model.overlap_query = False  # per-launch durations are only meaningful when the kernels run one at a time
    model.lite_overlap = model.lite_query_overlap = False  # (LITE: the H-subset and query passes otherwise run beside the cache pass)
    lib.orbit_prof_set_roofs(6.3e12, PEAK_FP32_MFMA_TFLOPS * 1e12, 64.0 * 1024.0 / 11.06e-9)
    lib.orbit_prof_enable(1)
    elapsed_prof, _, _ = loop(args.steps)
    lib.orbit_prof_enable(0)
    model.overlap_query, model.lite_overlap, model.lite_query_overlap = overlap, lite_overlap, lite_query_overlap
    prof_rows = collect_prof(lib)  # (every rank: the records are per process)
    per_rank = None
    if dist is not None:
        per_rank = per_rank_report(rank, world, dist, device, elapsed, issued, args.steps, run_step if train else None)
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(correct)  # frame-accuracy counts: the only exchange of the task-parallel form
        if train_losses is not None and train_losses.numel():
            train_losses = train_losses.to(device)
            dist.all_reduce(train_losses)
        torch.cuda.synchronize()

    rccl_ranks = rccl_check(lib, rank, world, backend, dist, device)  # every rank: it is a collective

    # The P2P / IPC self-check is the LAST collective of the run and runs under a watchdog: it exercises code no multi-GPU node
    # has executed yet (hipIpcOpenMemHandle across devices, peer writes over xGMI) and must not be able to cost the run its line -
    # every timed number and every other exchange is complete by now; if the check does not return within 90 s (or raises) the
    # line says so and the process ends with a hard exit instead of waiting for a communicator that may never drain.
    comm_check, comm_hung = None, False
    if dist is not None:
        import threading
        box = {}

        def guarded():
            try:
                torch.cuda.set_device(device)
                box["result"] = comm_selfcheck(rank, world, backend, dist, device)
            except BaseException as e:  # reported, never raised
                box["result"] = [{"rank": rank, "error": repr(e)[:300]}]
        th = threading.Thread(target=guarded, name="orbit-comm-selfcheck", daemon=True)
        th.start()
        th.join(float(os.environ.get("ORBIT_BENCH_SELFCHECK_TIMEOUT", "90")))
        comm_hung = th.is_alive()
        comm_check = [{"rank": rank, "error": "self-check did not return within the watchdog's limit"}] if comm_hung else box.get("result")

    def leave():
        sys.stdout.flush()
        if comm_hung:
            os._exit(0)  # (a stuck self-check thread holds the communicator: no orderly teardown)
        if dist is not None:
            dist.destroy_process_group()
    variants, other, total_bytes, n_main, conv_ms, conv_fl = conv_rows_summary(prof_rows)
    if rank != 0:
        leave()
        return

    ms, fl = ctypes.c_double(conv_ms), ctypes.c_double(conv_fl)
    achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process; the value is
    # the one tools/collect_profiles.sh measured with rocprofv3 on this same command (committed under profiles/)
    traffic, traffic_source = pmc_traffic(args.workload + (":lite_train" if train else ""))
    fam = family_table(prof_rows, args.steps * per_step, 1e3 * elapsed / args.steps / per_step)
    macs = model.feature_extractor.macs_per_frame(size, size)
    out = {
        "metric": baseline_metric(),
        "value": NUM_QUERY * args.steps * per_step * world / elapsed,
        "unit": "query frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "rccl_ranks": rccl_ranks, "ranks_share_gpus": bool(world > torch.cuda.device_count()),
        "per_rank": per_rank,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "weights": ("meta-trained with LITE through the native kernels (tools/meta_train.py), file %s" % os.path.relpath(ckpt, ROOT))
                   if ckpt else "deterministic random initialisation (synthetic.init_parameters_)",
        "task_family": template,
        "config": {"workload": "%s%s: ProtoNet + %s%s, %dx%d, %d-way, %d support frames (%d shots x %d), %d query "
                               "frames, clip_length 1, batch_size 256, inputs resident in HBM" % (
                                   args.workload,
                                   " LITE meta-training step (H=%d, fwd+bwd+fused Adam%s)" % (
                                       NUM_LITE, ", gradient all-reduce" if world > 1 else "") if train else "",
                                   fe_name, " + CNAPs FiLM adaptation" if adapt else "", size, size, way,
                                   WAY * SHOTS * FRAMES_PER_SHOT, SHOTS if way == WAY else 1,
                                   FRAMES_PER_SHOT if way == WAY else frames_per_class, NUM_QUERY),
                   "tasks_per_step": per_step * world,
                   "parallelism": "task-parallel x%d (independent tasks per rank%s)" % (
                       world, "; one all-reduce of the flat gradient bucket per optimizer step" if train and world > 1
                       else "")},
        "frame_accuracy": float(correct[0].item() / max(correct[1].item(), 1)),
        "train_graph_calls_replayed_eager": list(model.feature_extractor.train_graph_stats()) if train else None,
        # LITE: the H-clip subset pass of each task runs on a second stream beside the cache pass (ORBIT_LITE_OVERLAP=0: serial)
        "lite_subset_beside_cache_pass": bool(getattr(model, "lite_overlap", False)) if train else None,
        "lite_query_pass_beside_cache_pass": bool(getattr(model, "lite_query_overlap", False)) if train else None,
        "train_loss_per_step": [float(x) for x in train_losses.cpu()] if train_losses is not None else None,
        "median_task_ms": median_task_ms,
        "labels": "every timed task carries a label tensor new to the head: its label set is resolved inside the timed region "
                  "(orbit_label_set on a side stream, class count waited for at the head kernel)" if not train else
                  "every task's labels are uploaded from the host as a new tensor before the clock (unpack_task); their label set "
                  "comes from the host copy",
        # this rank's rate with the label sets of the resident tasks resolved and memoised before the clock (round 4's `value`)
        "value_memoised_labels": value_memoised,
        "value_overlap_off": value_overlap_off,
        # the library's default mode. Round 6: the recogniser's default IS the timed mode (overlap_query = "auto": the query pass
        # runs on the second stream when the clips are on the host or carry a readiness mark, which bench's resident clips do) -
        # one number; with ORBIT_BENCH_OVERLAP=2 (pipelined opt-in) the default mode is not what was timed and is not reported
        "value_default_mode": (NUM_QUERY * args.steps * per_step * world / elapsed) if timed_mode in ("auto", False) else None,
        "value_overlap_joined": value_overlap_joined,
        "overlap_mode": {False: "off", "auto": "library default (auto): query pass on a second stream from the clips' readiness "
                         "event, joined before the head", True: "query pass on a second stream, joined before the head", 2:
                         "pipelined: query pass and head on a second stream, not joined (tasks overlap out of phase)"}[
                             getattr(model, "overlap_query", False)],
        "host_enqueue_ms_per_step": 1e3 * issued / args.steps,
        # The label-set count of task k is produced right after task k-1's kernels (the stream the labels live on) and the
        # host needs it only at task k's head launch, after both extractor passes are queued: the host idles there until the
        # GPU reaches task k - it runs at most one task ahead, which costs nothing while its own work per task is shorter
        # than the GPU's. host work per step = host_enqueue_ms_per_step - host_label_wait_ms_per_step.
        "host_label_wait_ms_per_step": 1e3 * label_wait / args.steps,
        "graph_option": os.environ.get("ORBIT_GRAPH", "2 (adaptive)"),
        "settling_steps_before_warmup": settling,
        "overlap_query_stream": bool(getattr(model, "overlap_query", False)),
        "extractor_gflop_per_task": 2 * macs * (WAY * SHOTS * FRAMES_PER_SHOT + NUM_QUERY) / 1e9,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": total_bytes / max(n_main, 1),
                     "kernel": "the dense-convolution MFMA kernels: orbit::conv_igemm_kernel (all instantiations, incl. the reduce "
                               "pass of split-K launches) and orbit::pw_rgemm_kernel (pointwise register GEMM)" + (
                         " + orbit::conv_wgrad_kernel" if train else ""),
                     "launches": n_main, "avg_launch_us": 1e3 * ms.value / max(n_main, 1),
                     "algorithmic_hbm_gbs": total_bytes / (ms.value * 1e-3) / 1e9 if ms.value > 0 else None,
                     # share of the instrumented repeat's own wall time (overlap off there, so the kernels run one at a
                     # time and the share cannot exceed 1; round 3 divided by the overlapped loop's time)
                     "kernel_time_share": ms.value / (1e3 * elapsed_prof),
                     # the whole path against the same roof: every extractor FLOP of the task / the timed step / peak
                     "whole_task_frac": (2 * macs * (WAY * SHOTS * FRAMES_PER_SHOT + NUM_QUERY) * per_step * world * args.steps
                                         / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS) if not train else None,
                     "measured": "per-launch HIP events on the launch stream over a repeat of the %d timed steps with the "
                                 "support/query overlap switched off, so every kernel runs alone (instrumented repeat took "
                                 "%.1f ms/step)" % (args.steps, 1e3 * elapsed_prof / args.steps),
                     "variants": variants,  # (the kernels outside this family: `families` below)
                     # round 6: every kernel family of the task against its own floor
                     "families": fam["families"], "task_kernel_ms_serial": fam["task_kernel_ms_serial"],
                     "task_floor_ms": fam["task_floor_ms"], "whole_task_frac_of_floor": fam["whole_task_frac_of_floor"],
                     "task_floor_simd_ms": fam["task_floor_simd_ms"],
                     "whole_task_frac_of_floor_simd": fam["whole_task_frac_of_floor_simd"],
                     "floor_definition": fam["floor_definition"]},
        # N > 1: which carrier each exchange uses and whether the peer-to-peer inbox path works on this node (one-time check)
        "multi_gpu": None if dist is None else {
            "data_path_collective": ("one all-reduce(SUM) of the flat gradient bucket per optimizer step through %s" % (
                "the direct reduce-scatter + all-gather over P2P inboxes (ORBIT_BENCH_P2P_GRADIENTS=1)"
                if getattr(getattr(run_step, "bucket", None), "p2p", None) is not None else
                "torch.distributed all_reduce (backend %s%s)" % (backend, " = RCCL ring over xGMI" if backend == "nccl" else "")))
            if train else "none (task-parallel inference: independent tasks per rank; frame-accuracy counts all-reduced after "
                          "the clock)",
            "c_abi_rccl_allreduce_over_ranks": rccl_ranks, "comm_selfcheck": comm_check},
    }
    out["head_roofline"] = head_roofline(device)
    if not train and world == 1:
        out["variants_of_the_metric"] = metric_variants(model, tasks, device, min(args.steps, 20))
    bf3 = None
    if not train and world == 1 and os.environ.get("ORBIT_BENCH_BF3", "0") == "1":  # (frozen opt-in path: off unless asked for)
        # OPT-IN alternative, never `value`: the 14x14 / 7x7 pointwise convs with both operands split three
        # ways into bf16 and six products per fp32 product on the bf16 matrix cores (csrc/conv_bf3.hip, option conv_bf3). The
        # same timed loop, then the serial leg with per-launch events; its logit error against the pinned oracle is added
        # below from the cpu_baseline task.
        bf3_prev = lib.orbit_get_option(b"conv_bf3")  # (0 unless ORBIT_CONV_BF3 was set for an A/B run)
        lib.orbit_set_option(b"conv_bf3", 3)
        try:
            loop(3)
            el3, _, _ = loop(args.steps)
            ov = getattr(model, "overlap_query", False)
            model.overlap_query = False
            loop(2)
            off3, _, _ = loop(args.steps)
            lib.orbit_prof_enable(1)
            loop(args.steps)
            lib.orbit_prof_enable(0)
            model.overlap_query = ov
            ms3, fl3, n3 = ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
            lib.orbit_prof_collect(ctypes.byref(ms3), ctypes.byref(fl3), ctypes.byref(n3))
            split_ms = split_fl = conv_ms3 = conv_fl3 = 0.0
            for i in range(lib.orbit_prof_num_variants()):
                nm = ctypes.create_string_buffer(48)
                vms, vfl = ctypes.c_double(), ctypes.c_double()
                lib.orbit_prof_variant(i, nm, None, ctypes.byref(vms), ctypes.byref(vfl), None)
                if nm.value.decode().startswith("conv"):  # as `roofline`: the dense conv kernels, not the fused fronts
                    conv_ms3 += vms.value
                    conv_fl3 += vfl.value
                if nm.value.decode().startswith("conv_bf3<"):
                    split_ms += vms.value
                    split_fl += vfl.value
            ms3.value, fl3.value = conv_ms3, conv_fl3
            bf3 = {"option": "conv_bf3 = 3 (default 0): dense convs + the expand stage of the row-streaming fused fronts", "ms_per_step": 1e3 * el3 / args.steps,
                   "query_frames_per_s": NUM_QUERY * args.steps * per_step / el3,
                   "query_frames_per_s_overlap_off": NUM_QUERY * args.steps * per_step / off3,
                   "conv_tflops_all_dense_conv_launches": fl3.value / ms3.value / 1e9 if ms3.value else None,
                   "conv_frac_of_fp32_mfma_peak": fl3.value / ms3.value / 1e9 / PEAK_FP32_MFMA_TFLOPS if ms3.value else None,
                   "split_kernel_share_of_conv_flops": split_fl / fl3.value if fl3.value else None,
                   "split_kernel_tflops": split_fl / split_ms / 1e9 if split_ms else None,
                   "note": "fp32 operands split x = x0 + x1 + x2 (bf16, round to nearest, 24 significand bits), products "
                           "x0w0 + x0w1 + x1w0 + x1w1 + x0w2 + x2w0 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the fraction "
                           "is quoted against the fp32-MFMA peak the default path is priced on (FLOPs of the fp32 problem)"}
        except Exception as e:  # the experimental leg must never cost the default path its line
            bf3 = None
            out["opt_in_conv_bf3_error"] = repr(e)[:300]
        finally:
            lib.orbit_prof_enable(0)
            lib.orbit_set_option(b"conv_bf3", bf3_prev)
    out["opt_in_conv_bf3"] = bf3
    out["lite_train"] = None
    if not train and world == 1 and not HEADS.get(args.workload) and os.environ.get("ORBIT_BENCH_LITE_BLOCK", "1") != "0":
        try:
            out["lite_train"] = lite_train_block(args, device, lib, tasks, ckpt)
        except Exception as e:  # the extra leg must never cost the headline its line
            out["lite_train"] = {"error": repr(e)[:300]}
        finally:
            lib.orbit_prof_enable(0)
    if not args.no_cpu_baseline and world == 1:
        sd_before = {k: v.clone() for k, v in model.state_dict().items()} if train else None
        base, task, want = cpu_baseline(args.workload, model, train=train, way=way, template=template)
        if train:  # the same LITE step on the same weights and permutation
            import numpy as np
            model.load_state_dict(sd_before)
            np.random.seed(1991)
        got = run_step(model, {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}).cpu()
        base["max_abs_dlogit_vs_gpu"] = float((got - want).abs().max().item())
        base["argmax_identical"] = bool(torch.equal(got.argmax(1), want.argmax(1)))
        # the accuracy half of the metric on this task, both sides (utils/eval_metrics.py:27-36)
        base["frame_accuracy_gpu"] = float((got.argmax(1) == task["target_labels"]).float().mean())
        base["frame_accuracy_oracle"] = float((want.argmax(1) == task["target_labels"]).float().mean())
        out["cpu_baseline"] = base
        if bf3 is not None:  # the opt-in path's logits on the same task against the same oracle logits
            lib.orbit_set_option(b"conv_bf3", 3)
            try:
                got3 = run_step(model, {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}).cpu()
            finally:
                lib.orbit_set_option(b"conv_bf3", bf3_prev)
            # the same task twice more under the stream overlap: the path must be repeatable bit for bit (an earlier form of the
            # kernel was not - an LDS return overwrote operands of queued bf16 MFMAs when another stream's kernels were on the chip)
            lib.orbit_set_option(b"conv_bf3", 3)
            try:
                dev_task = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}
                bf3["bitwise_repeatable_under_overlap"] = all(bool(torch.equal(run_step(model, dev_task).cpu(), got3)) for _ in range(3))
            finally:
                lib.orbit_set_option(b"conv_bf3", bf3_prev)
            bf3["max_abs_dlogit_vs_oracle"] = float((got3 - want).abs().max().item())
            bf3["argmax_identical_to_oracle"] = bool(torch.equal(got3.argmax(1), want.argmax(1)))
            bf3["max_abs_dlogit_vs_default_path"] = float((got3 - got).abs().max().item())
            bf3["default_path_max_abs_dlogit_vs_oracle"] = base["max_abs_dlogit_vs_gpu"]
        # parity gate (no timing is reported for a path that does not reproduce the reference's logits)
        if not (base["max_abs_dlogit_vs_gpu"] <= 1e-3 and base["argmax_identical"]):
            print("PARITY GATE FAILED, no timing reported: " + json.dumps(base), file=sys.stderr)
            raise SystemExit(3)
        # the BASELINE configs this command does not time get one full-size HIP-vs-oracle task each in the
        # default run's record - config 4 (CNAPs FiLM adaptation + resnet18 @224) and config 5's task shape (10-way
        # efficientnet_b0 @224) - under the same gate. (ORBIT_BENCH_FULL_GATES=0 skips them: A/B runs.)
        gates = []
        if (not train and args.workload == "efficientnet_b0_224" and way == WAY
                and os.environ.get("ORBIT_BENCH_FULL_GATES", "1") != "0"):
            gates.append(dict(full_size_gate("efficientnet_b0_224", 10, device, base["cores"], template, model=model),
                              config="BASELINE configs[4] task shape: ProtoNet + efficientnet_b0, 224x224, 10-way"))
            gates.append(dict(full_size_gate("cnaps_resnet18_224", WAY, device, base["cores"], "noise"),
                              config="BASELINE configs[3]: CNAPs (set encoder + FiLM generator) + resnet18, 224x224, 5-way"))
            gates.append({"workload": args.workload, "way": way, "config": "BASELINE configs[2] (this run's cpu_baseline task)",
                          "max_abs_dlogit_vs_oracle": base["max_abs_dlogit_vs_gpu"], "argmax_identical": base["argmax_identical"]})
            for g in gates:
                if not (g["max_abs_dlogit_vs_oracle"] <= 1e-3 and g["argmax_identical"]):
                    print("PARITY GATE FAILED (full-size task of %s), no timing reported: %s" % (g["config"], json.dumps(g)),
                          file=sys.stderr)
                    raise SystemExit(3)
        out["full_size_parity"] = gates or None
    else:
        out["cpu_baseline"] = None
    out["parity_gate"] = gate
    print(json.dumps(out))
    leave()


if __name__ == "__main__":
    main()
