"""FiLM parameter generator on the native runtime.

Mirror of reference model/feature_adapters.py:36-95. Parameter names and initialisation are the
reference's (`generators.{i}` = DenseBlock(pooled, hidden, out_i), `regularizers.{i}` ~ N(0, 0.001), FiLM
names sorted); `forward(z)` returns the same `{name: tensor}` dict and sets `l2_term`, but all generators
run in one grouped HIP launch that writes the per-task BatchNorm gamma'/beta' straight into two
concatenated vectors laid out in the extractor's FiLM-slot order (the dict values are views into them, and
`last_film` holds the pair for the extractor's fast path).
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .mlps import DenseBlock


class FilmParameterGenerator(nn.Module):
    def __init__(self, film_parameter_sizes, initial_film_parameters, pooled_size, hidden_size,
                 slot_names=None):
        super().__init__()
        self.initial_film_parameters = initial_film_parameters
        self.film_parameter_names = sorted(self.initial_film_parameters.keys())
        self.pooled_size, self.hidden_size = pooled_size, hidden_size
        self.generators = nn.ModuleList()
        self.regularizers = nn.ParameterList()
        for name in self.film_parameter_names:
            out = film_parameter_sizes[name]
            self.generators.append(DenseBlock(pooled_size, hidden_size, out))
            self.regularizers.append(nn.Parameter(nn.init.normal_(torch.empty(out), 0, 0.001), requires_grad=True))
        self.l2_term = 0.0
        # destination layout: FiLM slots in the extractor's module order; 'x.weight' -> gamma, 'x.bias' -> beta
        if slot_names is None:
            slot_names = []
            for name in self.film_parameter_names:
                mod = name.rsplit(".", 1)[0]
                if mod not in slot_names:
                    slot_names.append(mod)
        self._slot_offset, off = {}, 0
        for mod in slot_names:
            self._slot_offset[mod] = off
            off += film_parameter_sizes[mod + ".weight"]
        self.film_size = off
        self._sizes = [film_parameter_sizes[n] for n in self.film_parameter_names]
        self._kinds = [0 if n.endswith(".weight") else 1 for n in self.film_parameter_names]
        self._dsts = [self._slot_offset[n.rsplit(".", 1)[0]] for n in self.film_parameter_names]
        self._handle = None
        self._stamp = None
        self.last_film = None

    def _apply(self, fn):  # keeps initial_film_parameters on the module's device (reference :55-58)
        super()._apply(fn)
        self.initial_film_parameters = {k: fn(v) for k, v in self.initial_film_parameters.items()}
        return self

    def regularization_term(self):
        return self.l2_term

    # ---- native handle ---------------------------------------------------------------------------
    def _tensors(self, i):
        blk = self.generators[i].block
        name = self.film_parameter_names[i]
        return (("w1", blk[0].weight), ("b1", blk[0].bias), ("ln_w", blk[1].weight), ("ln_b", blk[1].bias),
                ("w2", blk[3].weight), ("b2", blk[3].bias), ("reg", self.regularizers[i]),
                ("init", self.initial_film_parameters[name]))

    def _sync(self):
        lib = _lib.load()
        n = len(self.film_parameter_names)
        if self._handle is None:
            arr = ctypes.c_int * n
            h = ctypes.c_void_p()
            _lib.check(lib.orbit_filmgen_create(n, self.pooled_size, self.hidden_size, arr(*self._sizes),
                                                arr(*self._kinds), arr(*self._dsts), ctypes.byref(h)),
                       "orbit_filmgen_create")
            self._handle = h
        stamp = tuple((t.data_ptr(), t._version) for i in range(n) for _, t in self._tensors(i))
        if stamp != self._stamp:
            tensors = [t.detach() for i in range(n) for _, t in self._tensors(i)]
            if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
                # parameters resident on the device (they follow optimizer steps): one stream-ordered gather kernel on
                # the caller's stream instead of 8 x n blocking copies on the null stream
                ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
                _lib.check(lib.orbit_filmgen_load_all_async(self._handle, ptrs, len(tensors), _lib.stream_handle()),
                           "orbit_filmgen_load_all_async")
            else:
                for i in range(n):
                    for tname, t in self._tensors(i):
                        t = t.detach().contiguous().float()
                        _lib.check(lib.orbit_filmgen_load(self._handle, i, tname.encode(),
                                                          ctypes.c_void_p(t.data_ptr()), t.numel()),
                                   "orbit_filmgen_load")
            self._stamp = stamp

    def forward(self, x):
        _lib.require_gpu()
        self._sync()
        if x.numel() != self.pooled_size:
            raise ValueError("task embedding must have %d elements" % self.pooled_size)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_with_grad(x)
        z = x.detach().reshape(-1).contiguous().float()
        gamma = torch.empty(self.film_size, device=z.device, dtype=torch.float32)
        beta = torch.empty(self.film_size, device=z.device, dtype=torch.float32)
        l2 = torch.empty(1, device=z.device, dtype=torch.float32)
        _lib.check(_lib.load().orbit_filmgen_forward(self._handle, _lib.dptr(z, torch.float32), _lib.dptr(gamma),
                                                     _lib.dptr(beta), _lib.dptr(l2), _lib.stream_handle()),
                   "orbit_filmgen_forward")
        self.l2_term = l2[0]
        self.last_film = (gamma, beta)
        return self._film_dict(gamma, beta)

    def _film_dict(self, gamma, beta):
        film_dict = {}
        for name, size, kind, dst in zip(self.film_parameter_names, self._sizes, self._kinds, self._dsts):
            film_dict[name] = (gamma if kind == 0 else beta)[dst:dst + size]
        return film_dict

    def _forward_with_grad(self, x):
        """Meta-training form: one autograd node whose backward is orbit_filmgen_backward."""
        from .autograd import FilmGeneratorFunction
        lib = _lib.load()
        params, index = [], []
        for i in range(len(self.film_parameter_names)):
            for tname, t in self._tensors(i):
                if tname == "init":
                    continue
                params.append(t)
                index.append((lib.orbit_filmgen_param_offset(self._handle, i, tname.encode()), tuple(t.shape)))
        gamma, beta, l2 = FilmGeneratorFunction.apply(self, x, tuple(index), *params)
        self.l2_term = l2[0]
        self.last_film = (gamma, beta)
        return self._film_dict(gamma, beta)

    def __del__(self):
        try:
            if self._handle:
                _lib.load().orbit_filmgen_destroy(self._handle)
        except Exception:
            pass


class NullGenerator(nn.Module):
    """FiLM generator stand-in when adapt_features is False (reference :80-95)."""

    last_film = None

    def forward(self, x):
        return {}

    def regularization_term(self):
        return 0
