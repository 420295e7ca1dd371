"""Which path is closer to the exact answer? The headline task (efficientnet_b0 @224, 200 + 200 frames, 5-way ProtoNet) through
(a) the PyTorch-CPU oracle in fp32 (the parity target), (b) the same oracle in fp64 (the exact answer to ~1e-12), (c) the default
GPU path (fp32 MFMA), (d) the opt-in conv_bf3 path. Prints max |logit difference| for every pair against (b) and (a), the logit
scale, and the same for the pooled features' effect (argmax). A measurement tool for DESIGN.md, not part of the product.
Usage (GPU box): python tools/bf3_logit_error.py [frames_per_set, default 200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orbit_dataset_amd import _lib, synthetic
from oracle.recogniser import OracleRecogniser


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    torch.manual_seed(0)
    lib = _lib.load()
    device = torch.device("cuda", 0)
    trained = len(sys.argv) > 2 and sys.argv[2] == "trained"  # the meta-trained checkpoint + the "blobs" task family of bench.py
    model = bench.build_model("efficientnet_b0_224", device)
    if trained:
        bench.load_trained_checkpoint(model, bench.trained_checkpoint("efficientnet_b0_224"))
    ref = OracleRecogniser("efficientnet_b0", False, "proto", 1, 256, num_lite_samples=bench.NUM_LITE)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref.fe.load_state_dict({k[len("feature_extractor."):]: v for k, v in sd.items() if k.startswith("feature_extractor.")})
    task = synthetic.make_task(0, 5, 1, n // 5, n, 224, template="blobs" if trained else "noise")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref.personalise(task["context_clips"], task["context_labels"])
        l32 = ref.predict(task["target_clips"]).double()
        ref.fe.double()
        ref.personalise(task["context_clips"].double(), task["context_labels"])
        l64 = ref.predict(task["target_clips"].double())
        dev = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in task.items()}
        out = {}
        for opt in (0, 3):
            lib.orbit_set_option(b"conv_bf3", opt)
            out[1 if opt else 0] = bench.run_task(model, dev).double().cpu()
        lib.orbit_set_option(b"conv_bf3", 0)
    d = lambda a, b: (a - b).abs().max().item()
    print("%s weights, frames per set %d, |logit| max %.1f, mean %.1f" % ("meta-trained" if trained else "synthetic", n, l64.abs().max().item(), l64.abs().mean().item()))
    print("max |logit - fp64 oracle|:  fp32 oracle %.3e   GPU default (fp32 MFMA) %.3e   GPU conv_bf3 %.3e" % (d(l32, l64), d(out[0], l64), d(out[1], l64)))
    print("max |logit - fp32 oracle|:  GPU default %.3e   GPU conv_bf3 %.3e" % (d(out[0], l32), d(out[1], l32)))
    print("argmax equal to the fp64 oracle's:  fp32 oracle %s  default %s  conv_bf3 %s" % tuple(
        bool(torch.equal(x.argmax(1), l64.argmax(1))) for x in (l32, out[0], out[1])))


if __name__ == "__main__":
    main()
