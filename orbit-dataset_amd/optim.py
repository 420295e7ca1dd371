"""Loss and optimizer set-up of the learners (mirror of reference utils/optim.py:8-33)."""
import torch


def cross_entropy(test_logits, test_labels, reduction="mean"):
    """reference utils/optim.py:8-9 (`F.cross_entropy(test_logits, test_labels, reduction=reduction)`): forward and
    backward are native launches (csrc/loss.hip) on the logits' stream; there is no CPU form."""
    from .model.autograd import CrossEntropyFunction
    if test_logits.dim() != 2 or test_labels.dim() != 1 or test_labels.size(0) != test_logits.size(0):
        raise ValueError("cross_entropy expects logits [N, C] and labels [N] (got %s, %s)"
                         % (tuple(test_logits.shape), tuple(test_labels.shape)))
    if reduction not in ("none", "mean", "sum"):
        raise ValueError("%s is not a valid value for reduction" % reduction)
    if not test_logits.is_cuda:
        raise RuntimeError("orbit_dataset_amd.optim.cross_entropy runs on the GPU (HIP kernels); the logits are on %s"
                           % test_logits.device)
    return CrossEntropyFunction.apply(test_logits, test_labels.to(test_logits.device), reduction)


def mark_parameters_changed(model):
    """Tell the native plans that parameter VALUES changed without their tensor version counters moving.

    The plans re-upload parameters when a (data_ptr, _version) stamp changes. In-place ops bump `_version`; fused
    optimizers (`torch.optim.Adam(fused=True)`, `torch._fused_adam_`) do not — after such an update call this (the
    optimizer built by `init_optimizer` does it from a step hook)."""
    for m in model.modules():
        plans = m.__dict__.get("_plans")
        if plans:
            for pl in plans.values():
                pl.stamp = None
        if "_stamp" in m.__dict__ and not callable(m.__dict__["_stamp"]):
            m.__dict__["_stamp"] = None


def init_optimizer(model, lr, optimizer_type, args=None, extractor_lr_scale=0.1):
    """Parameter groups of reference utils/optim.py:11-33: everything but the extractor / the extractor. As in the
    reference the second group only carries the TAG `lr_scale`: it is timm's scheduler that multiplies it into the
    group's lr on every update (single-step learner; `apply_lr_scale` below does the same), so where no scheduler runs
    — the finetuner's personalise(), few_shot_recognisers.py:224 — the extractor trains at the full learning rate."""
    extractor_ids = set(map(id, model.feature_extractor.parameters()))
    base_params = [p for p in model.parameters() if id(p) not in extractor_ids]
    groups = [{"params": base_params},
              {"params": list(model.feature_extractor.parameters()), "lr_scale": extractor_lr_scale}]
    if optimizer_type == "adam":
        # args.fused_optimizer = True selects torch's fused multi-tensor Adam (one kernel per group and step instead of
        # ~10 foreach launches: 8 -> <1 ms of host time for efficientnet_b0's 213 tensors; the LITE step is GPU-bound, so
        # the step time does not move - measured - and the default stays the reference's plain Adam)
        fused = bool(getattr(args, "fused_optimizer", False))
        opt = torch.optim.Adam(groups, lr=lr, eps=getattr(args, "epsilon", 1e-8),
                               weight_decay=getattr(args, "weight_decay", 0.0),
                               betas=tuple(getattr(args, "betas", (0.9, 0.999))), fused=fused)
        if fused:  # the fused kernel does not bump the parameters' version counters
            opt.register_step_post_hook(lambda *_: mark_parameters_changed(model))
    elif optimizer_type == "sgd":
        opt = torch.optim.SGD(groups, lr=lr, momentum=getattr(args, "momentum", 0.0),
                              weight_decay=getattr(args, "weight_decay", 0.0))
    else:
        raise ValueError("optimizer %s not valid" % optimizer_type)
    opt.zero_grad()
    return opt


def apply_lr_scale(optimizer, lr):
    """What timm's Scheduler.update_groups does on every step: group lr = value * group['lr_scale']."""
    for group in optimizer.param_groups:
        group["lr"] = lr * group.get("lr_scale", 1.0)
