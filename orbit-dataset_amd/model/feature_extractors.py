"""Feature extractors backed by the native gfx950 runtime (csrc/extractor.hip).

Mirror of the reference factory `create_feature_extractor` (reference model/feature_extractors.py:37-79):
same signature and return value `(extractor, film_parameter_names)`, `extractor.output_size`, frozen
parameters when `learn_extractor=False`, FiLM tagging when `with_film=True`. The reference builds timm
networks; here the network lives in liborbit_hip.so and this module is only its parameter container
(state_dict-compatible key names: torchvision layout for resnet18, timm `tf_efficientnet_b0` layout for
efficientnet_b0) plus the call into `orbit_extractor_forward`. `resnet18` and arbitrary frame sizes are
additions BASELINE.json's configs require (the snapshot's args.py:27,77 no longer list them).

There is no pretrained-weight download (no network): `pretrained=True` is accepted for signature parity and
ignored; parameters are initialised deterministically by `synthetic.init_extractor_` or `load_state_dict`.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .autograd import ExtractorFunction

_EXTRACTOR_OUTPUT = {"resnet18": 512, "efficientnet_b0": 1280}
_BN_EPS = {"resnet18": 1e-5, "efficientnet_b0": 1e-3, "set_encoder": 1e-5}


class ParamNode(nn.Module):
    """Plain container node of the parameter tree (a conv, a BatchNorm, a block ...)."""

    def forward(self, *a, **k):  # pragma: no cover - never called
        raise RuntimeError("parameter container; the computation runs in liborbit_hip.so")


class BatchNormNode(ParamNode):
    """weight / bias parameters + running_mean / running_var / num_batches_tracked buffers of one BatchNorm."""


def _ensure_child(module, name, cls=ParamNode):
    child = module._modules.get(name)
    if child is None:
        child = cls()
        module.add_module(name, child)
    return child


class _Plan:
    """One native plan (frame size specific) and the parameter stamp it was last synchronised with."""

    def __init__(self, name, H, W, trainable=False):
        lib = _lib.load()
        h = ctypes.c_void_p()
        # the fused MBConv front kernels have no backward form: a plan that will record a tape is built without them
        _lib.check(lib.orbit_extractor_create_ex(name.encode(), H, W, 1 if trainable else 0, ctypes.byref(h)),
                   "orbit_extractor_create_ex")
        self.handle = h
        self.stamp = None
        self.generation = 0  # bumped by every parameter upload: a tape recorded under generation g can only be replayed under g
        self.workspaces = {}

    def destroy(self):
        if self.handle:
            _lib.load().orbit_extractor_destroy(self.handle)
            self.handle = None


class HipNetwork(nn.Module):
    """Parameter tree of a native network + forward through liborbit_hip.so.

    forward(frames[B,3,H,W] fp32 on the HIP device, film=None) -> features [B, output_size].
    `film` is an optional pair (gamma, beta) of concatenated per-task BatchNorm weight/bias for the FiLM slots
    (fast path). When the module is instead run under `torch.func.functional_call` with a FiLM dict (the
    reference's mechanism, few_shot_recognisers.py:114-115), the swapped-in BatchNorm tensors are detected and
    gathered automatically.

    Like any nn.Module the network follows `self.training`: in eval() BatchNorm uses running statistics (the
    inference runtime, fused kernels); in train() it uses batch statistics and updates the running statistics
    (the training runtime). With autograd enabled and a parameter / FiLM vector requiring a gradient, the forward
    records a tape and `backward()` runs the native gradient kernels (model/autograd.py).
    """

    def __init__(self, native_name):
        super().__init__()
        self.native_name = native_name
        self._plans = {}
        self._defer_stats = None  # list of (plan, tape, B) inside a deferred_stats block
        # persistent tape / feature buffers of taped forwards issued on a side stream (persistent_buffers below)
        self._persist_key = None
        self._persist = {}       # (key, kind) -> tensor
        self._persist_busy = {}  # key -> True while a tape recorded into the buffers awaits its backward
        lib = _lib.load()
        # a throw-away plan at a nominal size enumerates the state_dict keys and FiLM slots
        probe = _Plan(native_name, 64, 64)
        try:
            h = probe.handle
            self.output_size = lib.orbit_extractor_output_size(h)
            self._keys = []
            for i in range(lib.orbit_extractor_num_params(h)):
                key = lib.orbit_extractor_param_name(h, i).decode()
                numel = lib.orbit_extractor_param_numel(h, i)
                self._keys.append((key, numel))
            self._film_slot_names = []
            self._film_slot_channels = []
            for s in range(lib.orbit_extractor_film_slots(h)):
                self._film_slot_names.append(lib.orbit_extractor_film_slot_name(h, s).decode())
                self._film_slot_channels.append(lib.orbit_extractor_film_slot_channels(h, s))
            self.film_size = lib.orbit_extractor_film_size(h)
        finally:
            probe.destroy()
        self._leaves = []  # (module, attr, key)
        for key, numel in self._keys:
            self._register_leaf(key, numel)

    # ---- parameter tree -------------------------------------------------------------------------
    def _register_leaf(self, key, numel):
        parts = key.split(".")
        node = self
        is_bn_stat = parts[-1] in ("running_mean", "running_var")
        for i, part in enumerate(parts[:-1]):
            last = i == len(parts) - 2
            node = _ensure_child(node, part, BatchNormNode if (last and is_bn_stat) else ParamNode)
        attr = parts[-1]
        shape = self._leaf_shape(key, numel)
        if is_bn_stat:
            init = torch.zeros(shape) if attr == "running_mean" else torch.ones(shape)
            node.register_buffer(attr, init)
            if "num_batches_tracked" not in node._buffers:
                node.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
            self._leaves.append((node, attr, key, None))
        else:
            param = nn.Parameter(torch.zeros(shape))
            node.register_parameter(attr, param)
            # keep the Parameter object: functional_call swaps module._parameters entries, and a plan created
            # while FiLM tensors are swapped in must still upload the network's own BatchNorm parameters
            self._leaves.append((node, attr, key, param))

    def _leaf_shape(self, key, numel):
        return (numel,)

    def _tensor(self, node, attr):
        t = node._parameters.get(attr)
        return t if t is not None else node._buffers.get(attr)

    # ---- native plan handling ---------------------------------------------------------------------
    def _plan(self, H, W, trainable=False):
        """inference plans may hold fused ops that have no backward form; a forward that records a tape or uses batch
        statistics gets its own (unfused) plan. Both enumerate the same parameters in the same order."""
        plan = self._plans.get((H, W, trainable))
        if plan is None:
            plan = _Plan(self.native_name, H, W, trainable)
            lib = _lib.load()
            names = [lib.orbit_extractor_param_name(plan.handle, i).decode()
                     for i in range(lib.orbit_extractor_num_params(plan.handle))]
            if names != [k for k, _ in self._keys]:
                raise _lib.OrbitHipError("native plan enumerates different parameters than the module tree")
            self._plans[(H, W, trainable)] = plan
        return plan

    def in_sync(self, H, W, trainable=True):
        """True when the plan a forward at this frame size would use exists and holds the current parameters."""
        plan = self._plans.get((H, W, trainable))
        return plan is not None and plan.stamp == self._stamp()

    class persistent_buffers:
        """A taped forward issued inside the block records into buffers the network keeps under `key` (tape and features)
        instead of fresh allocations: a forward that runs on a side stream NOT ordered behind the caller's stream must not
        be handed an allocator block whose previous user is still in flight there, and fixed addresses let the training
        entry points replay their graphs. One tape per key at a time: `release(key)` (or the tape's backward) frees it;
        while it is busy `available(key)` is False and the caller takes its ordinary path."""

        def __init__(self, net, key):
            self.net, self.key = net, key

        def __enter__(self):
            self.net._persist_key = self.key
            return self

        def __exit__(self, *exc):
            self.net._persist_key = None
            return False

    def persistent_available(self, key):
        return not self._persist_busy.get(key, False)

    def persistent_release(self, key=None):
        if key is None:
            self._persist_busy.clear()
        else:
            self._persist_busy[key] = False

    def _persistent_tensor(self, key, kind, numel, dtype, device):
        t = self._persist.get((key, kind))
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != device:
            t = self._persist[(key, kind)] = torch.empty(max(int(numel), 256), dtype=dtype, device=device)
        return t[:numel] if t.numel() != numel else t

    def prepare(self, H, W, trainable=True):
        """Build (if needed) and bring up to date the plan a forward at this frame size will use, on the current stream."""
        self.sync(self._plan(H, W, trainable))

    def _stamp(self):
        # swapped-in FiLM tensors (functional_call) are plain tensors, not Parameters: they do not count as a
        # change of the network's own parameters
        st = []
        for node, attr, _, own in self._leaves:
            t = self._tensor(node, attr)
            if own is not None and not isinstance(t, nn.Parameter):
                t = own
            st.append((t.data_ptr(), t._version))
        return tuple(st)

    def sync(self, plan=None):
        """(Re)upload parameters into the native plan(s) if they changed since the last upload."""
        lib = _lib.load()
        stamp = self._stamp()
        plans = [plan] if plan is not None else list(self._plans.values())
        for pl in plans:
            if pl.stamp == stamp:
                continue
            tensors = []
            for node, attr, key, own in self._leaves:
                t = self._tensor(node, attr)
                if own is not None and not isinstance(t, nn.Parameter):
                    t = own
                tensors.append(t.detach())
            if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
                # every tensor already lives on the device: ONE gather kernel (the pointer table is cached inside the plan)
                # instead of one stream-ordered copy per tensor - after every optimizer step this was ~360 ctypes calls
                ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
                _lib.check(lib.orbit_extractor_load_all_async(pl.handle, ptrs, len(tensors), _lib.stream_handle()),
                           "orbit_extractor_load_all_async")
            else:
                for (node, attr, key, own), t in zip(self._leaves, tensors):
                    t = t.contiguous().float()
                    if t.is_cuda:  # stream-ordered copy: no host sync while parameters follow optimizer steps
                        _lib.check(lib.orbit_extractor_load_async(pl.handle, key.encode(), ctypes.c_void_p(t.data_ptr()),
                                                                  t.numel(), _lib.stream_handle()),
                                   "orbit_extractor_load_async(%s)" % key)
                    else:
                        _lib.check(lib.orbit_extractor_load(pl.handle, key.encode(), ctypes.c_void_p(t.data_ptr()),
                                                            t.numel()), "orbit_extractor_load(%s)" % key)
            _lib.check(lib.orbit_extractor_finalize(pl.handle, _lib.stream_handle()), "orbit_extractor_finalize")
            pl.stamp = stamp
            pl.generation += 1

    def _workspace(self, plan, B, device):
        # one workspace per (batch size, stream): forwards issued on different streams (the query pass overlapped with the
        # support pass, few_shot_recognisers.predict) must not share activation buffers
        key = (B, _lib.stream_handle().value or 0)
        ws = plan.workspaces.get(key)
        if ws is None or ws.device != device:
            nbytes = _lib.load().orbit_extractor_workspace_bytes(plan.handle, B)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            # keep the most recent batch size per stream
            plan.workspaces = {k: v for k, v in plan.workspaces.items() if k[1] != key[1]}
            plan.workspaces[key] = ws
        return ws

    def train_graph_stats(self):
        """(replayed, eager) calls of the native training entry points over this network's plans: the tape / gradient
        buffers come from torch's caching allocator, which hands back the same addresses step after step in a steady
        training loop - such calls replay a captured HIP graph instead of ~1 300 launches (option train_graph)."""
        lib = _lib.load()
        rep = eag = 0
        for pl in self._plans.values():
            a, b = ctypes.c_long(0), ctypes.c_long(0)
            lib.orbit_extractor_train_graph_stats(pl.handle, ctypes.byref(a), ctypes.byref(b))
            rep, eag = rep + a.value, eag + b.value
        return rep, eag

    def macs_per_frame(self, H, W):
        return _lib.load().orbit_extractor_macs_per_frame(self._plan(H, W).handle)

    # ---- FiLM ---------------------------------------------------------------------------------------
    def film_slot_modules(self):
        cached = self.__dict__.get("_film_slot_cache")
        if cached is None:
            mods = dict(self.named_modules())
            cached = [(name, mods[name]) for name in self._film_slot_names]
            self.__dict__["_film_slot_cache"] = cached
        return cached

    def _gather_swapped_film(self):
        """If BatchNorm weights/biases were swapped in by functional_call, return (gamma, beta) concatenated."""
        slots = self.film_slot_modules()
        if all(isinstance(m._parameters["weight"], nn.Parameter) and isinstance(m._parameters["bias"], nn.Parameter)
               for _, m in (slots[0], slots[-1])):
            return None  # functional_call with a FiLM dict swaps every slot; first and last suffice as a probe
        gammas = [m._parameters["weight"].detach().reshape(-1) for _, m in slots]
        betas = [m._parameters["bias"].detach().reshape(-1) for _, m in slots]
        return torch.cat(gammas).float().contiguous(), torch.cat(betas).float().contiguous()

    # ---- training path (tape + autograd) -------------------------------------------------------------
    bn_momentum = 0.1  # nn.BatchNorm2d default (torchvision resnet18, SimplePrePoolNet)

    def wants_grad(self, film=None):
        """True when a forward issued now has to record a tape: autograd is on and either one of the network's own
        parameters or the FiLM vectors require a gradient."""
        if not torch.is_grad_enabled():
            return False
        if film is not None and (film[0].requires_grad or film[1].requires_grad):
            return True
        return any(own is not None and own.requires_grad for _, _, _, own in self._leaves)

    def _param_index(self, plan):
        """[(flat-gradient offset, torch shape, is a FiLM-replaceable BatchNorm weight/bias)] per own Parameter."""
        cached = self.__dict__.get("_param_index_cache")
        if cached is None:
            lib = _lib.load()
            film_keys = set()
            for name in self._film_slot_names:
                film_keys.add(name + ".weight"), film_keys.add(name + ".bias")
            cached = []
            for i, (node, attr, key, own) in enumerate(self._leaves):
                if own is not None:
                    cached.append((own, (lib.orbit_extractor_param_offset(plan.handle, i), tuple(own.shape),
                                         key in film_keys)))
            self.__dict__["_param_index_cache"] = cached
        return cached

    def _pull_running_stats(self, plan):
        """Copy the running statistics a train-mode forward updated inside the plan back into the module's buffers
        (state_dict parity with nn.BatchNorm2d in train()), and bump num_batches_tracked."""
        lib = _lib.load()
        n = lib.orbit_extractor_bn_stat_floats(plan.handle)
        dev = None
        dst, src, counters = [], [], []
        off = 0
        for node, attr, key, own in self._leaves:
            if attr != "running_mean":
                continue
            rm, rv = node._buffers["running_mean"], node._buffers["running_var"]
            if dev is None:
                dev = rm.device
                flat = torch.empty(2, n, device=dev, dtype=torch.float32)
                _lib.check(lib.orbit_extractor_export_bn_stats(plan.handle, _lib.dptr(flat), _lib.stream_handle()),
                           "orbit_extractor_export_bn_stats")
            C = rm.numel()
            dst += [rm, rv]
            src += [flat[0, off:off + C], flat[1, off:off + C]]
            counters.append(node._buffers["num_batches_tracked"])
            off += (C + 3) // 4 * 4
        torch._foreach_copy_(dst, src)
        torch._foreach_add_(counters, 1)
        plan.stamp = self._stamp()  # the plan already holds these values

    def _forward_train(self, plan, x, film, use_tape, bn_train, out):
        lib = _lib.load()
        if not lib.orbit_extractor_supports_training(plan.handle):
            raise NotImplementedError(
                "this %s plan has no native training path (it was built with the fused MBConv front kernels); "
                "for inference call it in eval() under torch.no_grad()" % self.native_name)
        B = x.shape[0]
        gamma, beta = film if film is not None else (None, None)
        if use_tape:
            if out is not None:
                raise ValueError("`out=` cannot be combined with autograd")
            entries = [(own, meta) for own, meta in self._param_index(plan) if own.requires_grad]
            feats = ExtractorFunction.apply(self, plan, x, gamma, beta, bn_train, self.bn_momentum,
                                            tuple(meta for _, meta in entries), *[own for own, _ in entries])
        else:
            feats = out if out is not None else torch.empty(B, self.output_size, device=x.device, dtype=torch.float32)
            tape = torch.empty(lib.orbit_extractor_tape_bytes(plan.handle, B), dtype=torch.uint8, device=x.device)
            # no autograd node will ever read this tape (a cache pass under torch.no_grad()): ORBIT_TRAIN_NO_BACKWARD = 1
            # (+ 2 = ORBIT_TRAIN_DEFER_RUNNING_STATS inside a deferred_stats block)
            _lib.check(lib.orbit_extractor_train_forward_ex(
                plan.handle, _lib.dptr(x, torch.float32), B, _lib.dptr(gamma), _lib.dptr(beta), int(bn_train),
                float(self.bn_momentum), _lib.dptr(feats, torch.float32), ctypes.c_void_p(tape.data_ptr()), tape.numel(),
                1 | (2 if self._defer_stats is not None else 0), _lib.stream_handle()), "orbit_extractor_train_forward_ex")
            if self._defer_stats is not None:
                self._defer_stats.append((plan, tape, B))
        if bn_train and self._defer_stats is None:
            self._pull_running_stats(plan)
        return feats

    class deferred_stats:
        """Train-mode forwards issued inside the block leave the running statistics of the plan untouched (their batch
        statistics stay on the tape): such a forward may run on another stream (`_lib.use_stream`) beside a forward that
        does update them. `apply()` - after the streams have been joined - performs the updates in the order the forwards
        were issued and copies the result into the module's buffers, exactly as the forward itself would have."""

        def __init__(self, net):
            self.net, self.pending = net, []

        def __enter__(self):
            if self.net._defer_stats is not None:
                raise RuntimeError("deferred_stats blocks do not nest")
            self.net._defer_stats = self.pending
            return self

        def __exit__(self, *exc):
            self.net._defer_stats = None
            return False

        def apply(self):
            lib = _lib.load()
            for plan, tape, B in self.pending:
                _lib.check(lib.orbit_extractor_apply_deferred_bn_stats(
                    plan.handle, ctypes.c_void_p(tape.data_ptr()), tape.numel(), B, float(self.net.bn_momentum),
                    _lib.stream_handle()), "orbit_extractor_apply_deferred_bn_stats")
                self.net._pull_running_stats(plan)
            self.pending = []

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, film=None, out=None, check_sync=True):
        _lib.require_gpu()
        if x.dim() == 5:
            x = x.flatten(end_dim=1)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected frames of shape [B,3,H,W], got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise _lib.OrbitHipError("frames must be on the HIP device (got %s); no CPU fallback" % x.device)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        if film is None and self.film_size > 0:
            film = self._gather_swapped_film()
        if film is not None:
            if film[0].numel() != self.film_size or film[1].numel() != self.film_size:
                raise ValueError("film vectors must have %d elements" % self.film_size)
            film = (film[0].contiguous().float(), film[1].contiguous().float())
        use_tape = B > 0 and self.wants_grad(film)
        plan = self._plan(H, W, trainable=B > 0 and (use_tape or self.training))
        if check_sync or plan.stamp is None:
            self.sync(plan)
        if B > 0 and (use_tape or self.training):
            # batch-statistics BatchNorm and/or a recorded tape: the training runtime (csrc/extractor_train.hip)
            return self._forward_train(plan, x, film, use_tape, self.training, out)
        feats = out if out is not None else torch.empty(B, self.output_size, device=x.device, dtype=torch.float32)
        if B == 0:
            return feats
        ws = self._workspace(plan, B, x.device)
        gamma, beta = film if film is not None else (None, None)
        _lib.check(_lib.load().orbit_extractor_forward(
            plan.handle, _lib.dptr(x, torch.float32), B, _lib.dptr(gamma), _lib.dptr(beta),
            _lib.dptr(feats, torch.float32), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_handle()),
            "orbit_extractor_forward")
        return feats

    def __del__(self):
        try:
            for p in self._plans.values():
                p.destroy()
        except Exception:
            pass


# torch-layout shapes of the leaves, so that state_dicts interchange with torchvision / timm checkpoints
def _conv_shape(cout, cin, k):
    return (cout, cin, k, k)


class ResNet18(HipNetwork):
    def __init__(self):
        super().__init__("resnet18")

    def _leaf_shape(self, key, numel):
        if key == "conv1.weight":
            return (64, 3, 7, 7)
        if key.endswith("downsample.0.weight"):
            cout = {"layer2": 128, "layer3": 256, "layer4": 512}[key.split(".")[0]]
            return (cout, cout // 2, 1, 1)
        if ".conv" in key and key.endswith(".weight"):
            cout = {"layer1": 64, "layer2": 128, "layer3": 256, "layer4": 512}[key.split(".")[0]]
            cin = numel // (cout * 9)
            return (cout, cin, 3, 3)
        return (numel,)


class EfficientNetB0(HipNetwork):
    def __init__(self):
        super().__init__("efficientnet_b0")
        self._numel = dict(self._keys)

    def _leaf_shape(self, key, numel):
        if not key.endswith(".weight") or ".bn" in key or key.startswith("bn"):
            return (numel,)
        if key == "conv_stem.weight":
            return (32, 3, 3, 3)
        # the output-channel count equals the size of the tensor that follows a conv in module order:
        # derive it from the sibling BatchNorm / bias instead of hard-coding the table
        keys = dict(self._keys)
        prefix = key[: -len(".weight")]
        if prefix.endswith("conv_dw"):
            blk = prefix[: -len(".conv_dw")]
            bn = blk + (".bn1" if blk == "blocks.0.0" else ".bn2")
            c = keys[bn + ".weight"]
            k = int(round((numel // c) ** 0.5))
            return (c, 1, k, k)
        if prefix.endswith("se.conv_reduce") or prefix.endswith("se.conv_expand"):
            cout = keys[prefix + ".bias"]
            return (cout, numel // cout, 1, 1)
        if prefix == "conv_head":
            return (1280, numel // 1280, 1, 1)
        blk, conv = prefix.rsplit(".", 1)
        if blk == "blocks.0.0":
            bn = ".bn2"  # DepthwiseSeparableConv: conv_pw -> bn2
        else:
            bn = ".bn1" if conv == "conv_pw" else ".bn3"
        cout = keys[blk + bn + ".weight"]
        return (cout, numel // cout, 1, 1)


def create_feature_extractor(feature_extractor_name: str, pretrained: bool = True, with_film: bool = False,
                             learn_extractor: bool = True):
    """Same contract as the reference factory (model/feature_extractors.py:37-79)."""
    from .film import get_film_parameter_names, tag_film_layers

    if feature_extractor_name == "resnet18":
        feature_extractor = ResNet18()
    elif feature_extractor_name == "efficientnet_b0":
        feature_extractor = EfficientNetB0()
    else:
        raise ValueError(f"Invalid feature_extractor_name: {feature_extractor_name}")
    assert feature_extractor.output_size == _EXTRACTOR_OUTPUT[feature_extractor_name]

    if not learn_extractor:
        freeze_extractor(feature_extractor)

    film_param_names = None
    if with_film:
        tag_film_layers(feature_extractor_name, feature_extractor)
        film_param_names = get_film_parameter_names(feature_extractor_name, feature_extractor)
    return feature_extractor, film_param_names


def freeze_extractor(feature_extractor):
    for param in feature_extractor.parameters():
        param.requires_grad = False
