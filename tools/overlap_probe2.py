"""400 frames through efficientnet_b0@224 / resnet18@224 split over k concurrent streams (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
dev = torch.device("cuda:0")
for name, size in (("efficientnet_b0", 224), ("resnet18", 224), ("resnet18", 84)):
    fe, _ = create_feature_extractor(name, learn_extractor=False)
    synthetic.init_parameters_(fe); fe = fe.to(dev).eval()
    x = torch.randn(400, 3, size, size, device=dev)
    out = torch.empty(400, fe.output_size, device=dev)
    streams = [torch.cuda.Stream() for _ in range(8)]
    for k in (1, 2, 4, 8):
        n = 400 // k
        def run(reps):
            with torch.no_grad():
                for _ in range(reps):
                    for j in range(k):
                        with torch.cuda.stream(streams[j]):
                            fe(x[j * n:(j + 1) * n], out=out[j * n:(j + 1) * n], check_sync=False)
        run(3); torch.cuda.synchronize(); t0 = time.perf_counter(); run(10); torch.cuda.synchronize()
        print(name, size, "streams", k, "x", n, "frames: %.2f ms per 400 frames" % (1e3 * (time.perf_counter() - t0) / 10))
    n = 200
    def seq(reps):
        with torch.no_grad():
            for _ in range(reps):
                for j in range(2):
                    fe(x[j * n:(j + 1) * n], out=out[j * n:(j + 1) * n], check_sync=False)
    seq(3); torch.cuda.synchronize(); t0 = time.perf_counter(); seq(10); torch.cuda.synchronize()
    print(name, size, "sequential 2 x 200: %.2f ms" % (1e3 * (time.perf_counter() - t0) / 10))
