"""Op-level: one conv_bf3 layer on stream A, a loop of ONE other operator on stream B: which co-runner changes the conv's result?"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orbit_dataset_amd
from orbit_dataset_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
h = lambda s: ctypes.c_void_p(s.cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
B, H, Cin, Cout = 200, 14, 672, 112
x = torch.randn(B, H, H, Cin, device=dev, generator=g); w = torch.randn(Cout, Cin, 1, 1, device=dev, generator=g) / 26
gate = torch.rand(B, Cin, device=dev, generator=g)
def conv(y, s):
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(x), 0, _lib.dptr(w), _lib.dptr(y), None, None, None, _lib.dptr(gate), B, H, H, Cin, Cout, 1, 1, 1, 0, 0, H, H, 0, 0, h(s)))
# co-runners
xd = torch.randn(B, 14, 14, 672, device=dev, generator=g); wd = torch.randn(672, 1, 5, 5, device=dev, generator=g); yd = torch.empty_like(xd)
sc, sh = torch.ones(672, device=dev), torch.zeros(672, device=dev)
def dw5(s): _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(xd), _lib.dptr(wd), _lib.dptr(yd), _lib.dptr(sc), _lib.dptr(sh), B, 14, 14, 672, 5, 1, 2, 2, 14, 14, 2, h(s)))
xd7 = torch.randn(B, 7, 7, 1152, device=dev, generator=g); wd7 = torch.randn(1152, 1, 3, 3, device=dev, generator=g); yd7 = torch.empty_like(xd7)
sc7, sh7 = torch.ones(1152, device=dev), torch.zeros(1152, device=dev)
def dw3_7(s): _lib.check(lib.orbit_op_dwconv2d(_lib.dptr(xd7), _lib.dptr(wd7), _lib.dptr(yd7), _lib.dptr(sc7), _lib.dptr(sh7), B, 7, 7, 1152, 3, 1, 1, 1, 7, 7, 2, h(s)))
pooled = torch.randn(B, 672, device=dev, generator=g); w1 = torch.randn(28, 672, device=dev, generator=g); b1 = torch.zeros(28, device=dev)
w2 = torch.randn(672, 28, device=dev, generator=g); b2 = torch.zeros(672, device=dev); go = torch.empty(B, 672, device=dev)
def se(s): _lib.check(lib.orbit_op_se_gate(_lib.dptr(pooled), _lib.dptr(w1), _lib.dptr(b1), _lib.dptr(w2), _lib.dptr(b2), _lib.dptr(go), B, 672, 28, h(s)))
xe = torch.randn(B, 14, 14, 112, device=dev, generator=g); we = torch.randn(672, 112, 1, 1, device=dev, generator=g) / 10; ye = torch.empty(B, 14, 14, 672, device=dev)
def expand_fp32(s):
    lib.orbit_set_option(b"conv_bf3", 0)
    _lib.check(lib.orbit_op_conv2d(_lib.dptr(xe), 0, _lib.dptr(we), _lib.dptr(ye), None, None, None, None, B, 14, 14, 112, 672, 1, 1, 1, 0, 0, 14, 14, 2, 0, h(s)))
    lib.orbit_set_option(b"conv_bf3", 1)
xa = torch.randn(B, 7, 7, 320, device=dev, generator=g); ya = torch.empty(B, 320, device=dev)
def avg(s): _lib.check(lib.orbit_op_avgpool(_lib.dptr(xa), _lib.dptr(ya), B, 49, 320, h(s)))
def torch_add(s):
    with torch.cuda.stream(s):
        yd.add_(1.0)
lib.orbit_set_option(b"conv_bf3", 1)
ref = torch.empty(B, H, H, Cout, device=dev); conv(ref, sA); torch.cuda.synchronize()
for name, fn in (("nothing", None), ("dwconv 5x5 @14 x672 (dwconv_lds)", dw5), ("dwconv 3x3 @7 x1152", dw3_7), ("se_gate 672", se), ("fp32 expansion 112->672", expand_fp32),
                 ("avgpool", avg), ("torch elementwise add", torch_add)):
    bad, worst = 0, 0.0
    for rep in range(25):
        y = torch.empty(B, H, H, Cout, device=dev)
        torch.cuda.synchronize()
        if fn:
            for _ in range(6): fn(sB)
        conv(y, sA)
        if fn:
            for _ in range(6): fn(sB)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1; worst = max(worst, (y - ref).abs().max().item())
    print("co-runner %-36s: %2d of 25 conv outputs differ (max %.3e)" % (name, bad, worst), flush=True)
lib.orbit_set_option(b"conv_bf3", 0)
