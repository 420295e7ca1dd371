"""Phase timing of the LITE training step on the GPU box (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.learner import Learner, build_parser, cross_entropy
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 84
a = build_parser().parse_args(["--mode", "train", "--feature_extractor", "resnet18", "--learn_extractor", "--with_lite",
                               "--frame_size", str(fs), "--tasks_per_batch", "2"] + sys.argv[2:])
L = Learner(a); m = L.model; dev = L.device
from orbit_dataset_amd.learner import init_optimizer
opt = init_optimizer(m, a.learning_rate, a.optimizer, a, a.extractor_lr_scale)
m.set_test_mode(False); torch.set_grad_enabled(True)
import gc
if os.environ.get("GC") == "freeze": gc.collect(); gc.freeze()
if os.environ.get("GC") == "off": gc.disable()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
N = 16
pre = [L.make_train_task(i) for i in range(N)] if os.environ.get("PREGEN") else None
if os.environ.get("ONETHREAD"): torch.set_num_threads(1)
for i in range(N):
    task = pre[i] if pre else L.make_train_task(i)
    th = time.perf_counter()
    ctx, lab = task["context_clips"].to(dev), task["context_labels"].to(dev)
    tgt, tl = task["target_clips"].to(dev), task["target_labels"].to(dev)
    m._clear_caches()
    t0 = T(); m.personalise_with_lite(ctx, lab)
    t1 = T(); logits = m.predict_a_batch(tgt[:a.batch_size])
    t2 = T(); loss = len(lab) / (a.num_lite_samples * a.tasks_per_batch) * cross_entropy(logits, tl[:a.batch_size]); loss.backward()
    t3 = T(); m._reset()
    host_gap = 1e3 * (t0 - th)
    if i % 2 == 1: opt.step(); opt.zero_grad()
    t4 = T()
    print("task %d: personalise_with_lite %.1f ms | predict_a_batch %.1f | backward %.1f | opt %.1f | mem %.2f GB | h2d+sync before %.1f" % (i, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), torch.cuda.max_memory_allocated()/2**30, host_gap))
