import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orbit_dataset_amd
from orbit_dataset_amd import synthetic
from orbit_dataset_amd.model.classifier_heads import VersaClassifier, MahalanobisClassifier, PrototypicalClassifier
dev = torch.device('cuda:0')
for D in (512, 1280):
    feats = torch.randn(200, D, device=dev); lab = torch.arange(5, device=dev).repeat_interleave(40); q = torch.randn(200, D, device=dev)
    heads = {'proto': PrototypicalClassifier(1.0), 'versa': VersaClassifier(D, 1.0).to(dev), 'maha': MahalanobisClassifier(1.0)}
    for name, h in heads.items():
        for _ in range(3):
            h.configure(feats, lab); h.predict(q)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): h.configure(feats, lab)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20): h.predict(q)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(D, name, 'configure %.3f ms  predict %.3f ms' % (1e3*(t1-t0)/20, 1e3*(t2-t1)/20))
