"""Dataset-side input pipeline of the episodic path: JPEG directory -> decoded 8-bit frames -> pinned ring buffer ->
8-bit upload on a copy stream -> to_tensor + normalize on the GPU, double-buffered against the extractor.

What the reference does per task: `data/datasets.py:139-200` walks `root/<user>/<object>/<clean|clutter>/
<video>/*.jpg`, `:376-431` opens every frame with PIL, applies `to_tensor` + `normalize` on the host and returns fp32 clips
(602 KB per 224x224 frame); `data/queues.py:44,52-53` wraps that in a `DataLoader(pin_memory=False, num_workers=4/8)`, and
the recogniser moves each mini-batch to the device on the compute stream (`model/few_shot_recognisers.py:112,142`). Here:

  ORBITDirectory      the same directory walk (users -> objects -> clean / clutter -> videos -> sorted frames) and a test-task
                      sampler in the reference's layout (context = the clean videos of each object, target = its clutter
                      videos - or the clean ones left over, `datasets.py:153-161`), returning frame PATHS;
  decode_frames       PIL decode (as `datasets.py:422-431`) of a list of paths into one uint8 [n, H, W, 3] buffer, by a pool
                      of threads (Pillow releases the GIL while it decodes);
  TaskPrefetcher      a staging thread turns host tasks (8-bit frames, from the decoder or any other source) into device tasks:
                      pinned slot <- frames, uint8 H2D on a COPY stream (150 KB per frame instead of 602 KB), then
                      orbit_frames_from_uint8 (the reference transform, bit-identical, csrc/ingest.hip) on that stream into
                      the slot's fp32 clips, and an event the compute stream waits on. `depth` slots rotate, so the upload
                      and the normalisation of task i+1 run while the extractor works on task i.

`write_synthetic_orbit_directory` builds a small JPEG tree in that layout (the ORBIT dataset itself is not available
offline); it is test / benchmark scaffolding, not part of the path.
"""
import ctypes
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .utils import NORMALIZE_STATS


# ---- directory layout (reference data/datasets.py:139-200) -----------------------------------------------------------------
def write_synthetic_orbit_directory(root, users=2, objects_per_user=3, clean_videos=3, clutter_videos=2, frames_per_video=12,
                                    frame_size=224, seed=1991, quality=90):
    """root/<user>/<object>/<clean|clutter>/<video>/<video>-00001.jpg ...: every object has a colour / texture template
    (class signal), every video a smooth drift of it plus noise. Returns the number of frames written."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    n = 0
    for u in range(users):
        for o in range(objects_per_user):
            base = rng.randint(0, 256, size=(8, 8, 3)).astype(np.float32)
            for kind, count in (("clean", clean_videos), ("clutter", clutter_videos)):
                for v in range(count):
                    name = "P%03d--obj%02d--%s--%02d" % (u, o, kind, v)
                    d = os.path.join(root, "P%03d" % u, "obj%02d" % o, kind, name)
                    os.makedirs(d, exist_ok=True)
                    for f in range(frames_per_video):
                        small = np.clip(base + rng.normal(0, 20 if kind == "clean" else 45, base.shape) + 2.0 * f, 0, 255)
                        img = Image.fromarray(small.astype(np.uint8)).resize((frame_size, frame_size), Image.BILINEAR)
                        img.save(os.path.join(d, "%s-%05d.jpg" % (name, f + 1)), quality=quality)
                        n += 1
    return n


class ORBITDirectory:
    """The directory walk of reference data/datasets.py:139-200 (no annotation filters): users, their objects, the clean /
    clutter videos of every object and the sorted frame paths of every video."""

    def __init__(self, root, context_type="clean", target_type="clutter", min_context_frames=1, min_target_frames=1):
        self.root = root
        self.users, self.user2objs, self.obj2name, self.obj2vids, self.vid2frames = [], {}, {}, {}, {}
        obj_id = 0
        for user in sorted(os.listdir(root)):
            user_path = os.path.join(root, user)
            if not os.path.isdir(user_path):
                continue
            objs = []
            for obj_name in sorted(os.listdir(user_path)):
                obj_path = os.path.join(user_path, obj_name)
                clean_dir = os.path.join(obj_path, "clean")
                if not os.path.isdir(clean_dir):
                    continue
                clean = sorted(os.listdir(clean_dir))
                if context_type == "clean" and target_type == "clean":
                    split = min(5, len(clean) - 1)  # aim for 5 context videos, leaving at least 1 target video (:157-160)
                    sets = {"context": [("clean", v) for v in clean[:split]], "target": [("clean", v) for v in clean[split:]]}
                else:
                    clutter_dir = os.path.join(obj_path, "clutter")
                    clutter = sorted(os.listdir(clutter_dir)) if os.path.isdir(clutter_dir) else []
                    sets = {"context": [("clean", v) for v in clean], "target": [("clutter", v) for v in clutter]}
                kept = {"context": [], "target": []}
                for set_type, vids in sets.items():
                    need = min_context_frames if set_type == "context" else min_target_frames
                    for kind, v in vids:
                        vp = os.path.join(obj_path, kind, v)
                        frames = sorted(os.path.join(vp, f) for f in os.listdir(vp) if f.endswith(".jpg"))
                        if len(frames) >= need:
                            kept[set_type].append(vp)
                            self.vid2frames[vp] = frames
                if kept["context"] and kept["target"]:  # object is valid (:184-186)
                    objs.append(obj_id)
                    self.obj2name[obj_id] = obj_name
                    self.obj2vids[obj_id] = kept
                    obj_id += 1
            if objs:
                self.users.append(user)
                self.user2objs[user] = objs

    def user_task(self, user, clip_length=1, context_frames_per_video=None, target_frames_per_video=None):
        """One user-episodic task in the reference's task_dict layout (datasets.py:584-597) with PATHS instead of pixels:
        context_paths [N, T] / target_paths [M, T] (clips of T contiguous frames, non-overlapping for the context set, one
        clip ending at every frame - `attach_frame_history` - is the test loop's job for the target set), labels = the
        object's index within the user."""
        ctx, ctx_lab, tgt, tgt_lab, tgt_videos = [], [], [], [], []
        T = int(clip_length)
        for label, obj in enumerate(self.user2objs[user]):
            for vp in self.obj2vids[obj]["context"]:
                frames = self.vid2frames[vp][:context_frames_per_video]
                for i in range(0, len(frames) - T + 1, T):
                    ctx.append(frames[i:i + T])
                    ctx_lab.append(label)
            for vp in self.obj2vids[obj]["target"]:
                frames = self.vid2frames[vp][:target_frames_per_video]
                tgt_videos.append((len(tgt), len(tgt) + len(frames)))
                for f in frames:
                    tgt.append([f])
                    tgt_lab.append(label)
        return {"context_paths": np.array(ctx, dtype=object).reshape(len(ctx), T), "context_labels": torch.tensor(ctx_lab),
                "target_paths": np.array(tgt, dtype=object).reshape(len(tgt), 1), "target_labels": torch.tensor(tgt_lab),
                "target_videos": tgt_videos, "object_list": [self.obj2name[o] for o in self.user2objs[user]]}


def decode_frames(paths, out=None, pool=None):
    """JPEG files -> uint8 [n, H, W, 3] (RGB, as PIL decodes them: the input of to_tensor in datasets.py:428-429)."""
    from PIL import Image
    paths = list(paths)
    if not paths:  # a user without context / target frames: an empty batch, not an IndexError
        return out if out is not None else np.empty((0, 0, 0, 3), dtype=np.uint8)

    def one(i):
        with Image.open(paths[i]) as im:
            a = np.asarray(im.convert("RGB"))
        if out is not None:
            out[i].copy_(torch.from_numpy(a)) if isinstance(out, torch.Tensor) else out.__setitem__(i, a)
        return a

    if out is None:
        first = one(0)
        out = np.empty((len(paths),) + first.shape, dtype=np.uint8)
        out[0] = first
        rest = range(1, len(paths))
    else:
        rest = range(len(paths))
    if pool is None:
        for i in rest:
            one(i)
    else:
        list(pool.map(one, rest))
    return out


class DirectoryTaskSource:
    """Iterator of host tasks decoded from an ORBITDirectory: {context_clips u8 [N,T,H,W,3], context_labels, target_clips u8
    [M,1,H,W,3], target_labels, target_videos}. `workers` decode threads (the reference: DataLoader workers, queues.py:34)."""

    def __init__(self, directory, clip_length=1, workers=8, context_frames_per_video=None, target_frames_per_video=None,
                 users=None):
        self.dir, self.T = directory, int(clip_length)
        self.users = list(users if users is not None else directory.users)
        self.cpv, self.tpv = context_frames_per_video, target_frames_per_video
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))

    def __len__(self):
        return len(self.users)

    def __iter__(self):
        for user in self.users:
            t = self.dir.user_task(user, self.T, self.cpv, self.tpv)
            ctx = decode_frames(t["context_paths"].reshape(-1), pool=self.pool)
            tgt = decode_frames(t["target_paths"].reshape(-1), pool=self.pool)
            yield {"context_clips": torch.from_numpy(ctx).reshape(len(t["context_paths"]), self.T, *ctx.shape[1:]),
                   "context_labels": t["context_labels"],
                   "target_clips": torch.from_numpy(tgt).reshape(len(t["target_paths"]), 1, *tgt.shape[1:]),
                   "target_labels": t["target_labels"], "target_videos": t["target_videos"], "user": user}


class DatasetTaskSource:
    """Host tasks from a data/datasets.py dataset built with frames="uint8" (the reference-pinned sampler: way, video and clip
    sampling of reference data/datasets.py:289-336,433-469,540-598), in the layout TaskPrefetcher uploads:
    context_clips u8 [N,T,H,W,3]; a test-mode target set (a list of videos) is concatenated into target_clips u8 [M,1,H,W,3]
    with `target_videos` = [(lo, hi)] row ranges and one label per frame; a train-mode target set passes through as clips."""

    def __init__(self, dataset, indices=None):
        if dataset.frames != "uint8":
            raise ValueError("DatasetTaskSource needs a dataset built with frames='uint8'")
        self.dataset = dataset
        self.indices = list(range(len(dataset)) if indices is None else indices)

    def __len__(self):
        return len(self.indices)

    def __iter__(self):
        for i in self.indices:
            t = self.dataset[i]
            out = {"context_clips": t["context_clips"], "context_labels": t["context_labels"], "context_paths": t["context_paths"],
                   "object_list": t["object_list"], "task_id": t["task_id"], "user": t["task_id"]}
            if isinstance(t["target_clips"], list):
                ranges, lo = [], 0
                for frames in t["target_clips"]:
                    ranges.append((lo, lo + len(frames)))
                    lo += len(frames)
                out["target_clips"] = torch.cat(t["target_clips"]).unsqueeze(1)
                out["target_labels"] = torch.cat([lab.reshape(1).expand(hi - lo) for lab, (lo, hi) in
                                                  zip(t["target_labels"], ranges)]) if ranges else torch.empty(0, dtype=torch.int64)
                out["target_videos"], out["target_paths"] = ranges, t["target_paths"]
            else:
                out["target_clips"], out["target_labels"], out["target_paths"] = t["target_clips"], t["target_labels"], t["target_paths"]
            yield out


# ---- pinned ring + copy stream ---------------------------------------------------------------------------------------------
class _Slot:
    def __init__(self):
        self.pinned, self.dev_u8, self.dev_f32, self.dev_lab = {}, {}, {}, {}
        self.ready = None      # recorded on the copy stream when the slot's fp32 clips are complete
        self.released = None   # recorded on the consumer's stream when it is done with the slot
        self.task = None


class TaskPrefetcher:
    """Iterate over `source` (host tasks whose `*_clips` are uint8, channels last [..., H, W, 3] or channels first
    [..., 3, H, W] - or already-normalised float32 [..., 3, H, W], the reference's layout); yields the same dicts with the
    clips replaced by normalised fp32 [..., 3, H, W] tensors RESIDENT on `device`. Every other entry passes through (label
    tensors are moved to the device).

    A staging thread fills pinned slot buffers and issues, on a copy stream, the 8-bit upload and the normalisation kernel of
    task i+1 (and i+2 with depth 3) while the caller's stream runs the extractor on task i. The yielded tensors belong to the
    slot: they stay valid until the NEXT task is requested (then the slot is handed back: an event on the caller's stream
    makes the copy stream wait for the kernels that still read it)."""

    def __init__(self, source, device, depth=3, frame_norm_method="imagenet", consumer_streams=()):
        from .. import _lib
        _lib.require_gpu()
        self._lib = _lib
        self.source, self.device = source, torch.device(device)
        self.mean, self.std = NORMALIZE_STATS[frame_norm_method]
        self.depth = max(2, int(depth))
        self.slots = [_Slot() for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # further streams on which the consumer reads a slot's tensors (a recogniser in pipelined mode, overlap_query = 2, runs
        # the query pass on its own second stream and does not join it): a slot is only refilled once EVERY such stream has
        # passed the point at which the next task was requested. A callable is evaluated at release time.
        self.consumer_streams = consumer_streams
        self.free, self.full = queue.Queue(), queue.Queue()
        for s in self.slots:
            self.free.put(s)
        self.current = None
        self.error = None
        self.thread = threading.Thread(target=self._stage, name="orbit-task-prefetch", daemon=True)
        self.thread.start()

    @staticmethod
    def _buffer(store, key, shape, make):
        t = store.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n:
            t = store[key] = make(n)
        return t[:n].view(*shape)

    def _stage(self):
        try:
            torch.cuda.set_device(self.device)
            lib = self._lib.load()
            f3 = ctypes.c_float * 3
            mean, std = f3(*self.mean), f3(*self.std)
            for task in self.source:
                slot = self.free.get()
                if slot is None:
                    return
                for ev in slot.released or ():
                    self.copy_stream.wait_event(ev)  # the consumer's kernels that read the slot's last task
                out = dict(task)
                with torch.cuda.stream(self.copy_stream):
                    for key, val in task.items():
                        if not (isinstance(val, torch.Tensor) and key.endswith("clips")):
                            if isinstance(val, torch.Tensor) and key.endswith("labels"):
                                # labels live in the slot like the clips: the slot's `released` event orders their reuse
                                # after the consumer's kernels (a fresh allocation here would return to the COPY stream's
                                # pool when the consumer drops the task, while compute-stream kernels may still read it)
                                lab = self._buffer(slot.dev_lab, (key, val.dtype), val.shape,
                                                   lambda n, dt=val.dtype: torch.empty(n, dtype=dt, device=self.device))
                                lab.copy_(val, non_blocking=True)
                                out[key] = lab
                            continue
                        if val.dtype == torch.float32:
                            # already-normalised fp32 clips (the reference's own task_dict layout, data/datasets.py:584-597:
                            # 602 KB per 224x224 frame): no transform, but the upload of task i+1 still runs on the copy stream
                            # under the extractor's work on task i instead of in front of it
                            host = val
                            if not val.is_pinned():
                                host = self._buffer(slot.pinned, (key, "f32"), val.shape,
                                                    lambda n: torch.empty(n, dtype=torch.float32).pin_memory())
                                for ev in slot.released or ():
                                    ev.synchronize()
                                host.copy_(val)
                            f32 = self._buffer(slot.dev_f32, key, val.shape,
                                               lambda n: torch.empty(n, dtype=torch.float32, device=self.device))
                            f32.copy_(host, non_blocking=True)
                            out[key] = f32
                            continue
                        if val.dtype != torch.uint8:
                            raise ValueError("TaskPrefetcher: %s must be uint8 or float32 frames, got %s" % (key, val.dtype))
                        hwc = val.shape[-1] == 3 and val.shape[-3] != 3
                        *lead, a, b, c = val.shape
                        H, W = (a, b) if hwc else (b, c)
                        B = int(np.prod(lead)) if lead else 1
                        host = val
                        if not val.is_pinned():
                            host = self._buffer(slot.pinned, key, val.shape,
                                                lambda n: torch.empty(n, dtype=torch.uint8).pin_memory())
                            for ev in slot.released or ():
                                ev.synchronize()             # the previous upload from this pinned buffer has long finished;
                            host.copy_(val)                  # (host-side wait only matters if the consumer never advanced)
                        u8 = self._buffer(slot.dev_u8, key, val.shape,
                                          lambda n: torch.empty(n, dtype=torch.uint8, device=self.device))
                        f32 = self._buffer(slot.dev_f32, key, (*lead, 3, H, W),
                                           lambda n: torch.empty(n, dtype=torch.float32, device=self.device))
                        u8.copy_(host, non_blocking=True)
                        self._lib.check(lib.orbit_frames_from_uint8(self._lib.dptr(u8, torch.uint8), 1 if hwc else 0, B, H, W,
                                                                    mean, std, self._lib.dptr(f32), self._lib.stream_handle()),
                                        "orbit_frames_from_uint8")
                        out[key] = f32
                    slot.ready = torch.cuda.Event()
                    slot.ready.record(self.copy_stream)
                slot.task = out
                self.full.put(slot)
        except BaseException as e:  # surfaced by the consumer
            self.error = e
        finally:
            self.full.put(None)

    def _release_current(self):
        if self.current is not None:
            extra = self.consumer_streams() if callable(self.consumer_streams) else self.consumer_streams
            events = []
            for st in [torch.cuda.current_stream(self.device)] + [s for s in (extra or ()) if s is not None]:
                ev = torch.cuda.Event()
                ev.record(st)
                events.append(ev)
            self.current.released = events
            self.current.task = None
            self.free.put(self.current)
            self.current = None

    def __iter__(self):
        return self

    def __next__(self):
        self._release_current()
        slot = self.full.get()
        if slot is None:
            if self.error is not None:
                raise self.error
            raise StopIteration
        torch.cuda.current_stream(self.device).wait_event(slot.ready)
        # the clips were produced on the copy stream, complete at `slot.ready`: a recogniser in its default mode may start its
        # query pass from that event instead of behind the support pass (data.utils.mark_ready)
        from .utils import mark_ready
        for key, val in slot.task.items():
            if isinstance(val, torch.Tensor) and key.endswith("clips"):
                mark_ready(val, slot.ready)
        self.current = slot
        return slot.task

    def close(self):
        self._release_current()
        self.free.put(None)
