"""One 200-frame extractor forward on one stream against the same frames as 2 x 100 / 4 x 50 on internal streams (all ordered
after the caller's stream, joined back): what would a stream split INSIDE a call buy the default (non-overlapped) mode?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
dev = torch.device("cuda", 0)
for name in ("efficientnet_b0", "resnet18"):
    fe, _ = create_feature_extractor(name, True, False, False)
    synthetic.init_parameters_(fe)
    fe = fe.cuda().eval()
    x = torch.randn(200, 3, 224, 224, device=dev)
    out = torch.empty(200, fe.output_size, device=dev)
    streams = [torch.cuda.Stream() for _ in range(4)]
    main = torch.cuda.current_stream()
    def fwd(parts):
        if parts == 1:
            fe(x, out=out)
            return
        ev = torch.cuda.Event(); ev.record(main)
        n = 200 // parts
        for i in range(parts):
            s = streams[i]
            s.wait_event(ev)
            with torch.cuda.stream(s):
                fe(x[i * n:(i + 1) * n], out=out[i * n:(i + 1) * n])
            main.wait_stream(s)
    with torch.no_grad():
        ref = None
        for parts in (1, 2, 4, 1, 2, 4):
            for _ in range(5): fwd(parts)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): fwd(parts)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            if ref is None: ref = out.clone()
            print("%s: %d part(s) %.3f ms per 200 frames, identical to the single pass: %s" % (name, parts, 1e3 * dt, bool(torch.equal(out, ref))), flush=True)
