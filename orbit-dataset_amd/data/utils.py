"""Batch / clip utilities on the path of personalise() and predict().

Restatement of the three helpers of reference data/utils.py that the hot path's callers use:
  get_batch_indices    :49-54   [start, end) of mini-batch `index`, clipped to the last element
  attach_frame_history :8-28    sliding window of `history_length` frames per frame, left-padded with frame 0
  unpack_task          :30-47   task_dict -> tuple, labels moved to the device
"""
import torch


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
        context_labels = context_labels.to(device)
    if target_to_device and isinstance(target_labels, torch.Tensor):
        target_labels = target_labels.to(device)
    return (task_dict["context_clips"], task_dict.get("context_paths"), context_labels, task_dict["target_clips"],
            task_dict.get("target_paths"), target_labels, task_dict.get("object_list"))
