"""bench.py's output contract on a real GPU: ONE JSON line with BASELINE.json's metric on the headline workload, the
`roofline` block of the dominant kernel and the `cpu_baseline` block of the oracle, behind the parity gates."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the contract is about the LINE: the long legs of a real run are cut short (no settling chunks, one CPU-oracle task on two
# thread settings instead of the median of three on six)
FAST = dict(ORBIT_BENCH_SETTLE="0", ORBIT_BENCH_CPU_TASKS="1", ORBIT_BENCH_CPU_THREADS="8,16")


def _run(args, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=dict(os.environ, **FAST))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run(["--steps", "3", "--warmup", "1"])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "query frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "efficientnet_b0" in d["config"]["workload"] and "224x224" in d["config"]["workload"]
    assert "model" not in d["config"]
    # 200 query frames per task: value and ms_per_step describe the same measurement
    assert abs(d["value"] - 200.0 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0) and r["traffic_source"]
    assert d["parity_gate"]["max_abs_dfeature_vs_transformers"] <= d["parity_gate"]["tol"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["max_abs_dlogit_vs_gpu"] <= 1e-3 and cb["argmax_identical"] is True
    assert d["median_task_ms"] > 0 and d["value_overlap_off"] > 0
    # `value` is measured on tasks whose label tensors are new to the head; the memoised form is a separate field
    assert "new to the head" in d["labels"] and d["value_memoised_labels"] > 0
    assert "median of 1 tasks" in cb["sample"] and cb["cpu_model"]
    # the C-ABI's RCCL communicator ran (one rank here): all-reduce of ones == world
    assert d["rccl_ranks"] == 1 and d["ranks_share_gpus"] is False
    v = d["variants_of_the_metric"]
    assert v["h2d_inclusive_uint8_query_frames_per_s"] > 0 and v["h2d_inclusive_uint8_unpipelined_query_frames_per_s"] > 0
    # round 6: the timed mode IS the library's default mode (overlap_query = "auto" + clips marked ready): one number
    assert d["overlap_mode"].startswith("library default") and d["value_default_mode"] == d["value"]
    # ... every kernel family of the task against its own floor
    fams = {f["family"]: f for f in r["families"]}
    assert {"dense_conv", "fused_front", "stem", "depthwise", "se_gate", "head"} <= set(fams)
    for f in fams.values():
        assert f["us_per_task"] > 0 and f["launches_per_task"] > 0
    assert fams["dense_conv"]["x_floor"] > 1.0 and fams["fused_front"]["floor_simd_us_per_task"] > fams["fused_front"]["floor_us_per_task"]
    assert 0.0 < r["task_floor_ms"] < r["task_floor_simd_ms"] < r["task_kernel_ms_serial"]
    assert abs(r["whole_task_frac_of_floor"] - r["task_floor_ms"] / d["ms_per_step"]) < 1e-9
    # ... the LITE meta-training step in the same line (item 2), `value` untouched by it
    lt = d["lite_train"]
    assert "error" not in lt and lt["ms_per_step"] > 0 and lt["steps"] >= 10 and len(lt["train_loss_per_step"]) == lt["steps"]
    assert all(x > 0 for x in lt["train_loss_per_step"]) and 0.0 < lt["roofline"]["frac"] < 1.0
    assert lt["train_graph_calls_replayed_eager"][0] > 0
    # ... and one full-size HIP-vs-oracle task for every BASELINE config with a GPU (item 5)
    gates = d["full_size_parity"]
    assert len(gates) == 3 and {g["workload"] for g in gates} == {"efficientnet_b0_224", "cnaps_resnet18_224"}
    assert sorted(g["way"] for g in gates) == [5, 5, 10]
    assert all(g["max_abs_dlogit_vs_oracle"] <= 1e-3 and g["argmax_identical"] is True for g in gates)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (it used to
    run one rank and print n_gpus 1). Two ranks share this box's GPU through the gloo self-test backend; with the RCCL
    backend (one rank per GPU) the same command must refuse instead of printing a line with the wrong n_gpus."""
    env = dict(os.environ, ORBIT_BENCH_BACKEND="gloo", **FAST)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["tasks_per_step"] == 2 and d["ranks_share_gpus"] is True
    assert d["rccl_ranks"] is None  # RCCL refuses two ranks on one device: only checked one rank per GPU
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["ms_per_step"] > 0 for r in d["per_rank"])
    # round 6: the line says which carrier the data path uses and whether the P2P inbox path (HIP IPC) works on this node
    mg = d["multi_gpu"]
    assert mg["data_path_collective"].startswith("none") and [c["rank"] for c in mg["comm_selfcheck"]] == [0, 1]
    for c in mg["comm_selfcheck"]:
        assert c["p2p_ipc_open"].startswith("ok") and c["p2p_allreduce_ok"] is True and c["p2p_error_word"] == 0, c
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("ORBIT_BENCH_BACKEND")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                              "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_other_modes_run():
    d = _run(["--workload", "resnet18_84", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert "resnet18" in d["config"]["workload"] and d["value"] > 0 and "cpu_baseline" not in d or d["cpu_baseline"] is None
    t = _run(["--mode", "lite_train", "--workload", "resnet18_84", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert "LITE" in t["config"]["workload"] and t["ms_per_step"] > 0


def test_bench_config5_command_contract():
    """BASELINE config 5 is `bench.py --gpus 8 --mode lite_train --way 10 --tasks-per-rank 8`: 10-way efficientnet_b0 @224 LITE,
    64 tasks per optimizer step sharded over 8 ranks, one all-reduce of the flat gradient bucket per step (reference
    single-step-learner.py:162-166,231). Eight ranks share this box's one GPU through the gloo self-test backend; the line
    must say n_gpus 8 / tasks_per_step 64 and that the ranks shared a device (no scaling claim is made from it)."""
    env = dict(os.environ, ORBIT_BENCH_BACKEND="gloo", ORBIT_BENCH_SETTLE="0", OMP_NUM_THREADS="8", MKL_NUM_THREADS="8")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mode", "lite_train", "--way", "10",
                          "--tasks-per-rank", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["tasks_per_step"] == 64 and d["ranks_share_gpus"] is True
    assert "10-way" in d["config"]["workload"] and "LITE" in d["config"]["workload"] and "gradient all-reduce" in d["config"]["workload"]
    assert d["scaling"] == "weak" and d["ms_per_step"] > 0 and d["value"] > 0
    assert 0.0 <= d["frame_accuracy"] <= 1.0


def test_bench_training_form_two_ranks_reproduces_one_rank():
    """`bench.py --gpus 2 --mode lite_train` launches its own two ranks (gloo: they share this box's GPU), all-reduces the flat
    gradient bucket before every optimizer step and must reproduce the ONE-rank run that accumulates the same tasks (task i on
    rank i % 2; reference tasks_per_batch semantics, single-step-learner.py:162-166,231): the summed loss of every timed
    optimizer step agrees - the second step's only if the first step's all-reduced gradients moved the parameters alike. The
    line also carries the per-rank report a SCALE record needs (bucket bytes, all-reduce time by carrier)."""
    common = ["--mode", "lite_train", "--workload", "resnet18_84", "--steps", "3", "--warmup", "0", "--no-cpu-baseline",
              "--distinct-tasks", "6"]  # (tasks 0 .. 5 in windows of two on either run)
    env = dict(os.environ, ORBIT_BENCH_BACKEND="gloo", ORBIT_DIST_BACKEND="gloo", OMP_NUM_THREADS="8", **FAST)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    outs = []
    for extra in (["--gpus", "1", "--tasks-per-rank", "2"], ["--gpus", "2", "--tasks-per-rank", "1"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, capture_output=True, text=True,
                             timeout=900, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    one, two = outs
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["ranks_share_gpus"] is True and two["rccl_ranks"] is None
    assert one["config"]["tasks_per_step"] == two["config"]["tasks_per_step"] == 2
    la, lb = one["train_loss_per_step"], two["train_loss_per_step"]
    assert len(la) == len(lb) == 3 and all(x > 0 for x in la)
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-3 * abs(a), (la, lb)
    pr = two["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["gradient_bucket_bytes"] > 40e6 and r["allreduce_us_backend"] > 0  # resnet18: 11.2 M parameters
        assert r.get("allreduce_us_p2p", 0) > 0 or "allreduce_p2p_error" in r
