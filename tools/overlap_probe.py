"""How much do two independent extractor forwards gain from running on two streams? (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
dev = torch.device("cuda:0")
for name, size in (("efficientnet_b0", 224), ("resnet18", 84), ("resnet18", 224)):
    nets = []
    for i in range(2):
        fe, _ = create_feature_extractor(name, learn_extractor=False)
        synthetic.init_parameters_(fe)
        nets.append(fe.to(dev).eval())
    x = [torch.randn(200, 3, size, size, device=dev) for _ in range(2)]
    outs = [torch.empty(200, nets[0].output_size, device=dev) for _ in range(2)]
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    def seq(n):
        with torch.no_grad():
            for _ in range(n):
                with torch.cuda.stream(s[0]):
                    nets[0](x[0], out=outs[0]); nets[1](x[1], out=outs[1])
    def par(n):
        with torch.no_grad():
            for _ in range(n):
                with torch.cuda.stream(s[0]): nets[0](x[0], out=outs[0])
                with torch.cuda.stream(s[1]): nets[1](x[1], out=outs[1])
    for fn in (seq, par, seq, par):
        fn(3); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(10); torch.cuda.synchronize()
        print(name, size, fn.__name__, "%.2f ms per pair of forwards" % (1e3 * (time.perf_counter() - t0) / 10))
