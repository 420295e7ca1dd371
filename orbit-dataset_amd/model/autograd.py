"""torch.autograd bridges to the native training kernels (csrc/extractor_train.hip, train_ops.hip, film.hip).

The reference trains through plain PyTorch autograd (single-step-learner.py:196-243: `loss.backward()` after
`personalise[_with_lite]` + `predict[_a_batch]`). Here each native stage is one autograd node whose backward is one
C-ABI call; torch only routes the gradients between the nodes and into `param.grad` (so `torch.optim` and
`zero_grad` work unchanged, as in utils/optim.py:11-33).

  ExtractorFunction      orbit_extractor_train_forward / orbit_extractor_backward   (resnet18, efficientnet_b0, set encoder)
  ProtoPredictFunction   orbit_proto_predict / orbit_proto_predict_backward          (gradient w.r.t. query features;
                         the prototypes are constants, classifier_heads.py:261-263 re-wraps them in nn.Parameter)
  MahalanobisPredictFunction  orbit_mahalanobis_predict / orbit_mahalanobis_predict_backward (w.r.t. query features)
  FilmGeneratorFunction  orbit_filmgen_forward / orbit_filmgen_backward
  MeanPoolFunction       orbit_mean_pool (+ broadcast backward)
  CrossEntropyFunction   orbit_cross_entropy_forward / orbit_cross_entropy_backward  (the learners' loss, utils/optim.py:8-9)
  SetMeanFunction        orbit_set_mean  (+ broadcast backward)
"""
import ctypes

import torch

from .. import _lib


def _empty_bytes(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class ExtractorFunction(torch.autograd.Function):
    """feats = net(frames; film) with a tape. Inputs after `momentum` are the network's trainable Parameters (only so
    that autograd routes their gradients); their values are read from the native plan, which `net.sync` keeps equal."""

    @staticmethod
    def forward(ctx, net, plan, frames, gamma, beta, bn_train, momentum, param_index, *params):
        lib = _lib.load()
        B = frames.shape[0]
        dev = frames.device
        key = net._persist_key
        if key is not None and net.persistent_available(key):
            # (HipNetwork.persistent_buffers: a forward on a side stream records into buffers the network owns)
            tape = net._persistent_tensor(key, "tape", lib.orbit_extractor_tape_bytes(plan.handle, B), torch.uint8, dev)
            feats = net._persistent_tensor(key, "feats", B * net.output_size, torch.float32, dev).view(B, net.output_size)
            net._persist_busy[key] = True
        else:
            key = None
            tape = _empty_bytes(lib.orbit_extractor_tape_bytes(plan.handle, B), dev)
            feats = torch.empty(B, net.output_size, device=dev, dtype=torch.float32)
        ctx.persist_key = key
        # (flag 2 = ORBIT_TRAIN_DEFER_RUNNING_STATS: the network applies the running-statistics update later, see
        # HipNetwork.deferred_stats)
        _lib.check(lib.orbit_extractor_train_forward_ex(
            plan.handle, _lib.dptr(frames, torch.float32), B, _lib.dptr(gamma), _lib.dptr(beta), int(bn_train),
            float(momentum), _lib.dptr(feats), ctypes.c_void_p(tape.data_ptr()), tape.numel(),
            2 if net._defer_stats is not None else 0, _lib.stream_handle()), "orbit_extractor_train_forward")
        if net._defer_stats is not None:
            net._defer_stats.append((plan, tape, B))
        ctx.net, ctx.plan, ctx.tape, ctx.bn_train, ctx.param_index = net, plan, tape, int(bn_train), param_index
        ctx.save_for_backward(frames, gamma, beta)
        ctx.film_needs = (gamma is not None and gamma.requires_grad) or (beta is not None and beta.requires_grad)
        ctx.generation = plan.generation
        return feats

    @staticmethod
    def backward(ctx, dfeats):
        lib = _lib.load()
        frames, gamma, beta = ctx.saved_tensors
        net, plan = ctx.net, ctx.plan
        if plan.generation != ctx.generation:
            # the backward kernels read filters / BatchNorm weights from the native plan, not from the tape: a parameter
            # upload between forward and backward (optimizer step, load_state_dict, mark_parameters_changed + another
            # forward) would silently differentiate a different network than the one that ran forward
            raise RuntimeError("the extractor's parameters were modified (re-uploaded into the native plan) between the "
                               "forward that recorded this tape and its backward; run backward before changing them")
        B, dev = frames.shape[0], frames.device
        n_fixed = 8
        need_params = any(ctx.needs_input_grad[n_fixed:])
        need_film = gamma is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        out = [None] * (n_fixed + len(ctx.param_index))
        if not need_params and not need_film:
            return tuple(out)
        flat = torch.zeros(lib.orbit_extractor_grad_floats(plan.handle), device=dev) if need_params else None
        dgamma = torch.empty_like(gamma) if need_film else None
        dbeta = torch.empty_like(beta) if need_film else None
        ws = _empty_bytes(lib.orbit_extractor_backward_workspace_bytes(plan.handle, B), dev)
        dfeats = dfeats.contiguous().float()
        # only BatchNorm weights / biases trainable (FiLM fine-tuning of a frozen extractor): skip the filter gradients
        filter_grads = any(need and not is_bn for need, (_, _, is_bn) in zip(ctx.needs_input_grad[n_fixed:],
                                                                            ctx.param_index))
        _lib.check(lib.orbit_extractor_backward(
            plan.handle, _lib.dptr(frames, torch.float32), B, _lib.dptr(gamma), _lib.dptr(beta), ctx.bn_train,
            _lib.dptr(dfeats), ctypes.c_void_p(ctx.tape.data_ptr()), ctx.tape.numel(), _lib.dptr(flat),
            int(filter_grads), _lib.dptr(dgamma), _lib.dptr(dbeta), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
            _lib.stream_handle()),
            "orbit_extractor_backward")
        ctx.tape = None
        if ctx.persist_key is not None:
            net.persistent_release(ctx.persist_key)
        if need_film:
            out[3], out[4] = dgamma, dbeta
        if need_params:
            for j, (off, shape, replaced_by_film) in enumerate(ctx.param_index):
                if not ctx.needs_input_grad[n_fixed + j] or (replaced_by_film and gamma is not None):
                    continue
                n = 1
                for d in shape:
                    n *= d
                out[n_fixed + j] = flat[off:off + n].view(shape)
        return tuple(out)


class ProtoPredictFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, T, logit_scale, cosine):
        q = features.contiguous().float()
        MT, D = q.shape
        M, C = MT // T, weight.size(0)
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        if M > 0:
            _lib.check(_lib.load().orbit_proto_predict(
                _lib.dptr(q, torch.float32), _lib.dptr(weight), _lib.dptr(None if cosine else bias), 1, M, T, D, C,
                float(logit_scale), int(cosine), _lib.dptr(logits), _lib.dptr(None), _lib.stream_handle()),
                "orbit_proto_predict")
        ctx.save_for_backward(q, weight)
        ctx.args = (M, T, D, C, float(logit_scale), int(cosine))
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        q, weight = ctx.saved_tensors
        M, T, D, C, scale, cosine = ctx.args
        dq = torch.empty_like(q)
        if M > 0:
            dl = dlogits.contiguous().float()
            _lib.check(_lib.load().orbit_proto_predict_backward(
                _lib.dptr(dl), _lib.dptr(q), _lib.dptr(weight), M, T, D, C, scale, cosine, _lib.dptr(dq),
                _lib.stream_handle()), "orbit_proto_predict_backward")
        return dq, None, None, None, None, None


class LinearPredictFunction(torch.autograd.Function):
    """logits = s (q . W^T + b) with gradients for the features, W and b (the multi-step finetuner's head)."""

    @staticmethod
    def forward(ctx, features, weight, bias, logit_scale):
        q = features.contiguous().float()
        M, D = q.shape
        C = weight.size(0)
        w, b = weight.detach().contiguous().float(), bias.detach().contiguous().float()
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        if M > 0:
            _lib.check(_lib.load().orbit_proto_predict(_lib.dptr(q), _lib.dptr(w), _lib.dptr(b), 1, M, 1, D, C,
                                                       float(logit_scale), 0, _lib.dptr(logits), _lib.dptr(None),
                                                       _lib.stream_handle()), "orbit_proto_predict")
        ctx.save_for_backward(q, w)
        ctx.scale = float(logit_scale)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        q, w = ctx.saved_tensors
        M, D = q.shape
        C = w.size(0)
        dl = dlogits.contiguous().float()
        dq = dw = db = None
        if M > 0 and ctx.needs_input_grad[0]:
            dq = torch.empty_like(q)
            _lib.check(lib.orbit_proto_predict_backward(_lib.dptr(dl), _lib.dptr(q), _lib.dptr(w), M, 1, D, C, ctx.scale, 0,
                                                        _lib.dptr(dq), _lib.stream_handle()), "orbit_proto_predict_backward")
        if M > 0 and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            dw, db = torch.empty_like(w), torch.empty(C, device=q.device, dtype=torch.float32)
            _lib.check(lib.orbit_linear_head_backward(_lib.dptr(dl), _lib.dptr(q), M, D, C, ctx.scale, _lib.dptr(dw),
                                                      _lib.dptr(db), _lib.stream_handle()), "orbit_linear_head_backward")
        return dq, dw, db, None


class MahalanobisPredictFunction(torch.autograd.Function):
    """logits = -s (mu_c - q)^T P_c (mu_c - q); gradient w.r.t. the query features only (means / precisions are the
    re-wrapped constants of classifier_heads.py:324-327)."""

    @staticmethod
    def forward(ctx, features, means, precisions, logit_scale):
        lib = _lib.load()
        q = features.contiguous().float()
        M, D = q.shape
        C = means.size(0)
        logits = torch.empty(M, C, device=q.device, dtype=torch.float32)
        ws = _empty_bytes(lib.orbit_mahalanobis_workspace_bytes(2, M, D, C), q.device)
        _lib.check(lib.orbit_mahalanobis_predict(_lib.dptr(q), _lib.dptr(means), _lib.dptr(precisions), M, D, C,
                                                 float(logit_scale), _lib.dptr(logits), ctypes.c_void_p(ws.data_ptr()),
                                                 ws.numel(), _lib.stream_handle()), "orbit_mahalanobis_predict")
        ctx.save_for_backward(q, means, precisions)
        ctx.scale = float(logit_scale)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        q, means, precisions = ctx.saved_tensors
        M, D = q.shape
        C = means.size(0)
        dq = torch.empty_like(q)
        ws = _empty_bytes(lib.orbit_mahalanobis_workspace_bytes(2, M, D, C), q.device)
        dl = dlogits.contiguous().float()
        _lib.check(lib.orbit_mahalanobis_predict_backward(_lib.dptr(dl), _lib.dptr(q), _lib.dptr(means),
                                                          _lib.dptr(precisions), M, D, C, ctx.scale, _lib.dptr(dq),
                                                          ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                                          _lib.stream_handle()), "orbit_mahalanobis_predict_backward")
        return dq, None, None, None


class FilmGeneratorFunction(torch.autograd.Function):
    """(gamma', beta', l2) = generator(z). Inputs after `z` are the generator's Parameters in `param_index` order."""

    @staticmethod
    def forward(ctx, gen, z, param_index, *params):
        lib = _lib.load()
        zc = z.detach().reshape(-1).contiguous().float()
        gamma = torch.empty(gen.film_size, device=zc.device, dtype=torch.float32)
        beta = torch.empty(gen.film_size, device=zc.device, dtype=torch.float32)
        l2 = torch.empty(1, device=zc.device, dtype=torch.float32)
        _lib.check(lib.orbit_filmgen_forward(gen._handle, _lib.dptr(zc, torch.float32), _lib.dptr(gamma),
                                             _lib.dptr(beta), _lib.dptr(l2), _lib.stream_handle()),
                   "orbit_filmgen_forward")
        ctx.gen, ctx.param_index, ctx.z_shape = gen, param_index, z.shape
        ctx.save_for_backward(zc)
        return gamma, beta, l2

    @staticmethod
    def backward(ctx, dgamma, dbeta, dl2):
        lib = _lib.load()
        (zc,) = ctx.saved_tensors
        gen = ctx.gen
        dev = zc.device
        zeros = lambda g, n: torch.zeros(n, device=dev) if g is None else g.contiguous().float()
        dgamma, dbeta = zeros(dgamma, gen.film_size), zeros(dbeta, gen.film_size)
        dl2 = None if dl2 is None else dl2.contiguous().float()
        flat = torch.zeros(lib.orbit_filmgen_grad_floats(gen._handle), device=dev)
        dz = torch.empty_like(zc)
        _lib.check(lib.orbit_filmgen_backward(gen._handle, _lib.dptr(zc), _lib.dptr(dgamma), _lib.dptr(dbeta),
                                              _lib.dptr(dl2), _lib.dptr(flat), _lib.dptr(dz), _lib.stream_handle()),
                   "orbit_filmgen_backward")
        out = [None, dz.view(ctx.z_shape) if ctx.needs_input_grad[1] else None, None]
        for j, (off, shape) in enumerate(ctx.param_index):
            if not ctx.needs_input_grad[3 + j]:
                out.append(None)
                continue
            n = 1
            for d in shape:
                n *= d
            out.append(flat[off:off + n].view(shape))
        return tuple(out)


class MeanPoolFunction(torch.autograd.Function):
    """[n*T, D] -> [n, D] mean over the T frames of a clip (reference model/poolers.py:13-16)."""

    @staticmethod
    def forward(ctx, x, T):
        x = x.contiguous().float()
        D = x.size(-1)
        n = x.numel() // (T * D)
        out = torch.empty(n, D, device=x.device, dtype=torch.float32)
        if n > 0:
            _lib.check(_lib.load().orbit_mean_pool(_lib.dptr(x, torch.float32), n, T, D, _lib.dptr(out),
                                                   _lib.stream_handle()), "orbit_mean_pool")
        ctx.T = T
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.T).repeat_interleave(ctx.T, dim=0), None


class SetMeanFunction(torch.autograd.Function):
    """[n, D] -> [1, D] mean over the set (reference model/set_encoders.py:61-75, aggregation='mean')."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        out = torch.empty(1, x.shape[1], device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().orbit_set_mean(_lib.dptr(x, torch.float32), x.shape[0], x.shape[1], _lib.dptr(out),
                                              _lib.stream_handle()), "orbit_set_mean")
        ctx.n = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).expand(ctx.n, -1).contiguous()


_REDUCTIONS = {"none": 0, "mean": 1, "sum": 2}


class CrossEntropyFunction(torch.autograd.Function):
    """loss = F.cross_entropy(logits, labels, reduction) (reference utils/optim.py:8-9) as two native launches; the
    backward reads the upstream gradient on the device (no host synchronisation inside the training step)."""

    @staticmethod
    def forward(ctx, logits, labels, reduction):
        red = _REDUCTIONS[reduction]
        z = logits.contiguous().float()
        lab = labels.contiguous().long()
        N, C = z.shape
        row_loss = torch.empty(N, device=z.device, dtype=torch.float32)
        softmax = torch.empty(N, C, device=z.device, dtype=torch.float32)
        loss = torch.empty((), device=z.device, dtype=torch.float32)
        _lib.check(_lib.load().orbit_cross_entropy_forward(
            _lib.dptr(z), _lib.dptr(lab, torch.int64), N, C, red, _lib.dptr(row_loss), _lib.dptr(softmax),
            _lib.dptr(loss), _lib.stream_handle()), "orbit_cross_entropy_forward")
        ctx.save_for_backward(softmax, lab)
        ctx.args = (N, C, red)
        return row_loss if red == 0 else loss

    @staticmethod
    def backward(ctx, grad):
        softmax, lab = ctx.saved_tensors
        N, C, red = ctx.args
        dz = torch.empty_like(softmax)
        g = grad.contiguous().float()
        _lib.check(_lib.load().orbit_cross_entropy_backward(
            _lib.dptr(softmax), _lib.dptr(lab, torch.int64), _lib.dptr(g), N, C, red, _lib.dptr(dz),
            _lib.stream_handle()), "orbit_cross_entropy_backward")
        return dz, None, None
