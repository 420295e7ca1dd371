import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd  # noqa
from oracle import extractors as oe
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.model.feature_extractors import create_feature_extractor
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))
ref = oe.create("resnet18"); nat, _ = create_feature_extractor("resnet18", with_film=True, learn_extractor=True)
synthetic.init_parameters_(ref), synthetic.init_parameters_(nat)
ref = ref.double(); nat = nat.to(dev)
sd0 = {k: v.clone() for k, v in nat.state_dict().items()}; rsd0 = {k: v.clone() for k, v in ref.state_dict().items()}
for size, B, train in ((33, 4, False), (33, 4, True), (64, 8, True), (84, 4, False)):
    for seed in range(6):
        nat.load_state_dict(sd0); ref.load_state_dict(rsd0); nat.zero_grad(); ref.zero_grad()
        ref.train(train); nat.train(train)
        x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(1000 + seed))
        d = torch.randn(B, 512, generator=torch.Generator().manual_seed(seed))
        # count near-zero ReLU inputs in the fp64 oracle
        near = []
        hooks = []
        def hook(mod, inp, out):
            z = inp[0].detach()
            near.append(int((z.abs() < 1e-5).sum()))
        for m_ in ref.modules():
            if isinstance(m_, torch.nn.ReLU): hooks.append(m_.register_forward_hook(hook))
        o_ref = ref(x.double()); o_ref.backward(d.double())
        for h in hooks: h.remove()
        o = nat(x.to(dev)); o.backward(d.to(dev))
        rg = dict(ref.named_parameters())
        errs = [(n, rel(p.grad, rg[n].grad)) for n, p in nat.named_parameters()]
        bad = [n for n, e in errs if e > 1e-4]
        print(size, B, train, "seed", seed, "worst %.1e" % max(e for _, e in errs), "n_bad", len(bad), "deepest_bad", bad[-1] if bad else "-", "relu inputs |z|<1e-5:", sum(near))
