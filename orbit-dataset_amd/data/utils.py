"""Batch / clip utilities on the path of personalise() and predict().

Restatement of the three helpers of reference data/utils.py that the hot path's callers use:
  get_batch_indices    :49-54   [start, end) of mini-batch `index`, clipped to the last element
  attach_frame_history :8-28    sliding window of `history_length` frames per frame, left-padded with frame 0
  unpack_task          :30-47   task_dict -> tuple, labels moved to the device
plus the input-side counterpart of data/datasets.py:422-431 (`frames_from_uint8`): decoded 8-bit frames are uploaded
as they are (a quarter of the fp32 bytes over PCIe) and normalised on the GPU, bit-identically to to_tensor + normalize.
"""
import ctypes

import torch

NORMALIZE_STATS = {  # reference data/datasets.py:82-87
    "imagenet": ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]),
    "imagenet_inception": ([0.5, 0.5, 0.5], [0.5, 0.5, 0.5]),
    "openai_clip": ([0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]),
}


def frames_from_uint8(frames, device, frame_norm_method="imagenet", channels_last=True):
    """uint8 frames [..., H, W, 3] (channels_last, as decoded) or [..., 3, H, W] -> normalised fp32 [..., 3, H, W] on
    `device`. Host tensors are uploaded as 8-bit (pin them for an asynchronous copy); leading dimensions are kept."""
    from .. import _lib
    _lib.require_gpu()
    if frames.dtype != torch.uint8:
        raise ValueError("expected uint8 frames, got %s" % frames.dtype)
    mean, std = NORMALIZE_STATS[frame_norm_method]
    if channels_last:
        *lead, H, W, C = frames.shape
    else:
        *lead, C, H, W = frames.shape
    if C != 3:
        raise ValueError("expected 3 colour channels, got %d" % C)
    u8 = frames.to(device, non_blocking=True).contiguous()
    B = 1
    for d in lead:
        B *= d
    out = torch.empty(*lead, 3, H, W, device=u8.device, dtype=torch.float32)
    if B > 0:
        f3 = ctypes.c_float * 3
        _lib.check(_lib.load().orbit_frames_from_uint8(_lib.dptr(u8, torch.uint8), 1 if channels_last else 0, B, H, W,
                                                       f3(*mean), f3(*std), _lib.dptr(out), _lib.stream_handle()),
                   "orbit_frames_from_uint8")
    return out


def mark_ready(tensor, event=None):
    """Declare a device tensor READY as of now: every operation that produces its content has been queued on the current
    stream (or is covered by `event`, e.g. the copy-stream event of the upload that filled it). A recogniser in its default
    mode (`overlap_query = "auto"`) then starts the query pass of predict() on its second stream from THAT point instead of
    behind the support pass personalise() queued on the caller's stream: it no longer has to assume that the clips might be
    the product of work still pending there. `data/pipeline.TaskPrefetcher` marks the clips it yields; host-resident clips
    need no mark (their upload is issued on the second stream). The mark lives on the tensor OBJECT: views and slices of it
    are unmarked (and take the serial order) unless marked themselves. Returns the tensor."""
    if isinstance(tensor, torch.Tensor) and tensor.is_cuda:
        if event is None:
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(tensor.device))
        tensor._orbit_ready = event
    return tensor


def ready_event(tensor):
    """The readiness event `mark_ready` attached to this tensor object, or None."""
    return getattr(tensor, "_orbit_ready", None)


def get_batch_indices(index, last_element, batch_size):
    start = index * batch_size
    return start, min(start + batch_size, last_element)


def attach_frame_history(frames, history_length):
    """frames [F,3,H,W] -> clips [F, history_length, 3, H, W]; clip f = frames f-L+1 .. f (frame 0 repeated
    where the window reaches before the video start). Built with one gather instead of roll/stack."""
    L = int(history_length)
    F = frames.shape[0]
    if L <= 1:
        return frames.unsqueeze(1)
    idx = torch.arange(F, device=frames.device)[:, None] + torch.arange(-(L - 1), 1, device=frames.device)[None, :]
    return frames[idx.clamp_(min=0)]


def unpack_task(task_dict, device, context_to_device=True, target_to_device=False):
    context_labels = task_dict["context_labels"]
    target_labels = task_dict["target_labels"]
    if context_to_device and isinstance(context_labels, torch.Tensor):
        host_labels = context_labels
        context_labels = context_labels.to(device)
        if not host_labels.is_cuda and context_labels.is_cuda:
            # the labels arrive on the host (the reference moves them here, data/utils.py:42-43): the task's label set - what
            # the head's configure needs the COUNT of before it can shape anything - is taken from the host copy, so neither
            # a torch.unique on the device nor its host sync happens later (model/classifier_heads.py memoises it per tensor)
            from ..model.classifier_heads import PrototypicalClassifier
            PrototypicalClassifier.register_label_set(context_labels, host_labels)
    if target_to_device and isinstance(target_labels, torch.Tensor):
        target_labels = target_labels.to(device)
    return (task_dict["context_clips"], task_dict.get("context_paths"), context_labels, task_dict["target_clips"],
            task_dict.get("target_paths"), target_labels, task_dict.get("object_list"))
