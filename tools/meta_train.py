#!/usr/bin/env python
"""Meta-train the recogniser with LITE on class-separable synthetic tasks and write the checkpoint the bench / parity runs
load (the headline's frame accuracy was chance on randomly initialised weights).

    python tools/meta_train.py [--steps 300] [--frame_size 224] [--out orbit-dataset_amd/assets/meta_trained_<fe>_<size>.npz]

The reference's outer loop (single-step-learner.py:136-194): for every task train_task_with_lite (per query batch:
personalise_with_lite -> predict_a_batch -> N/(H*tasks_per_batch) * CE -> backward), optimizer.step() every tasks_per_batch
tasks - here through orbit_dataset_amd.learner.Learner, i.e. the native forward / backward kernels, on tasks of the "blobs"
family (synthetic.make_task, low-frequency class templates) drawn on the device. Afterwards the model is evaluated in test
mode (personalise + predict, running-statistics BatchNorm) on held-out tasks, before / after training.

The checkpoint is the state_dict rounded to fp16 (10.6 MB for efficientnet_b0): whoever loads it - the HIP path, the CPU
oracle - reads identical values, so parity statements are about these exact weights. A JSON log of the run (loss / accuracy
per step, held-out accuracy) goes next to it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import learner, synthetic  # noqa: E402


def evaluate(model, device, size, way, tasks, first_index):
    accs = []
    model.set_test_mode(True)
    with torch.no_grad():
        for i in range(tasks):
            t = synthetic.make_task_on_device(first_index + i, way, 1, 200 // way, 200, size, 1, device, template="blobs")
            model.personalise(t["context_clips"], t["context_labels"])
            logits = model.predict(t["target_clips"])
            accs.append(float((logits.argmax(1) == t["target_labels"]).float().mean()))
            model._reset()
    model.set_test_mode(False)
    return float(np.mean(accs)), accs


def save_checkpoint(model, path):
    """state_dict -> npz, floating tensors rounded to fp16 (and the model re-loaded from the rounded values, so the
    evaluation that follows describes exactly what the file holds)."""
    sd = {}
    for k, v in model.state_dict().items():
        v = v.detach().cpu()
        sd[k] = v.half().numpy() if v.is_floating_point() else v.numpy()
    np.savez_compressed(path, **sd)
    return load_checkpoint(model, path)


def load_checkpoint(model, path):
    sd = {k: torch.from_numpy(v.astype(np.float32) if v.dtype == np.float16 else v) for k, v in np.load(path).items()}
    model.load_state_dict(sd)
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feature_extractor", default="efficientnet_b0")
    ap.add_argument("--frame_size", type=int, default=224)
    ap.add_argument("--steps", type=int, default=300, help="training tasks")
    ap.add_argument("--tasks_per_batch", type=int, default=4)
    ap.add_argument("--learning_rate", type=float, default=1e-3)
    ap.add_argument("--way", type=int, default=5)
    ap.add_argument("--eval_tasks", type=int, default=10)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = a.out or os.path.join(ROOT, "orbit-dataset_amd", "assets",
                                "meta_trained_%s_%d.npz" % (a.feature_extractor, a.frame_size))
    args = learner.build_parser().parse_args([
        "--mode", "train", "--feature_extractor", a.feature_extractor, "--learn_extractor", "--with_lite",
        "--frame_size", str(a.frame_size), "--tasks_per_batch", str(a.tasks_per_batch), "--learning_rate",
        str(a.learning_rate), "--weight_decay", "0.0", "--fused_optimizer", "--way", str(a.way)])
    L = learner.Learner(args)
    model, device = L.model, L.device
    L.optimizer = learner.init_optimizer(model, args.learning_rate, args.optimizer, args, args.extractor_lr_scale)
    learner.apply_lr_scale(L.optimizer, args.learning_rate)
    log = {"args": vars(a), "steps": []}
    log["heldout_acc_before"], _ = evaluate(model, device, a.frame_size, a.way, a.eval_tasks, 900_000)
    print("held-out frame accuracy before training: %.3f" % log["heldout_acc_before"], flush=True)
    model.set_test_mode(False)
    t0 = time.perf_counter()
    with torch.enable_grad():
        for step in range(a.steps):
            task = synthetic.make_task_on_device(100_000 + step, a.way, 1, 200 // a.way, 200, a.frame_size, 1, device,
                                                 template="blobs")
            np.random.seed((args.seed + 7919 * (step + 1)) % (2 ** 32))
            loss, logits = L.train_task_with_lite(task)
            if (step + 1) % a.tasks_per_batch == 0 or step == a.steps - 1:
                L.optimizer.step()
                L.optimizer.zero_grad()
            acc = float((logits.argmax(1) == task["target_labels"]).float().mean())
            log["steps"].append({"step": step, "loss": float(loss) * a.tasks_per_batch, "frame_acc": acc})
            if step % 20 == 0 or step == a.steps - 1:
                print("step %4d  loss %.4f  train-task frame_acc %.3f" % (step, float(loss) * a.tasks_per_batch, acc), flush=True)
    torch.cuda.synchronize()
    log["train_seconds"] = time.perf_counter() - t0
    save_checkpoint(model, out)
    log["heldout_acc_after"], per_task = evaluate(model, device, a.frame_size, a.way, a.eval_tasks, 900_000)
    log["heldout_acc_after_per_task"] = per_task
    log["checkpoint"] = os.path.relpath(out, ROOT)
    log["checkpoint_bytes"] = os.path.getsize(out)
    first = np.mean([s["loss"] for s in log["steps"][:20]])
    last = np.mean([s["loss"] for s in log["steps"][-20:]])
    log["loss_first20"], log["loss_last20"] = float(first), float(last)
    print("trained %d tasks in %.1f s; loss %.4f -> %.4f; held-out frame accuracy %.3f -> %.3f; %s (%.1f MB)"
          % (a.steps, log["train_seconds"], first, last, log["heldout_acc_before"], log["heldout_acc_after"], out,
             log["checkpoint_bytes"] / 1e6), flush=True)
    with open(os.path.splitext(out)[0] + ".json", "w") as f:
        json.dump(log, f)


if __name__ == "__main__":
    main()
