"""Episodic task sampling over an ORBIT directory: the dataset side of the hot path's inputs.

Host-side mirror of reference data/datasets.py (same class names, constructor arguments and task_dict keys, so
`data/queues.py`-style callers and single-step-learner.py's loops can switch over), pinned to the reference by fixture G14
(tests/golden/make_golden.py imports the reference's own module with torchvision's two transforms stubbed and records what
its `__getitem__` returns on a committed JPEG tree; tests/test_datasets.py replays the same seeds through this module).

What is restated, with the reference lines it follows:

  index            :104-205   sorted walk root/<user>/<object>/<clean|clutter>/<video>/*.jpg; clean/clean splits the clean
                              videos 5 + rest (at least one target); a context video needs >= 1 frame, a target video >= 50;
                              an object needs both sets, a user needs an object; video ids count kept videos in walk order
  way              :289-301   'max' = min(#objects, object_cap); 'random' = one draw from [2, that]
  videos           :313-336   'specific' / 'fixed' / 'random' / 'max' under the shot cap (15; with_caps: 5 / 4 for >= 6-way
                              tasks, else 10 / 8, :548-551)
  clips            :433-469   frame ids capped at 1000, padded with the last frame to a multiple of clip_length, viewed as
                              non-overlapping clips; 'max' / 'random' / 'random_200' / 'uniform' (subsample_factor, clip cap 200)
  sets             :471-522   train: clips + labels shuffled together; test: grouped by video id (ascending), frames flattened
  frames           :422-431   PIL decode, to_tensor, normalize

Every random draw goes through ONE `random.Random`-compatible source in the reference's order (way, objects, then per object:
context videos, target videos, context clips per video, target clips per video; then the context shuffle, then the target
shuffle), so seeding the `random` module (the default source, as the reference uses it) reproduces the reference's tasks.

Beyond the reference: `frames=` selects what the clips hold - "float" (normalised fp32, the reference's layout), "uint8"
(decoded 8-bit [.., H, W, 3] for TaskPrefetcher: a quarter of the bytes over PCIe, normalised on the GPU bit-identically) or
"paths" (nothing decoded). Frame annotations and cluster labels are not on the recognition path and are not built:
asking for them raises.
"""
import glob
import os
import random as _random

import numpy as np
import torch

from .utils import NORMALIZE_STATS

FRAME_CAP = 1000      # frames considered per video (reference :80)
CLIP_CAP = 200        # clips sampled per video (reference :79)
MIN_FRAMES = {"context": 1, "target": 50}  # reference :121-134


def clip_frame_indices(num_frames, clip_length, method, subsample_factor=30, rng=_random):
    """Frame ids [n_clips * clip_length] of the clips sampled from one video (reference :433-469)."""
    ids = np.arange(min(int(num_frames), FRAME_CAP))
    tail = len(ids) % clip_length
    if tail:
        ids = np.concatenate([ids, np.full(clip_length - tail, ids[-1])])
    n = len(ids) // clip_length
    as_clips = ids.reshape(n, clip_length)
    if method == "max":
        pick = as_clips
    elif method == "random":
        count = rng.choice(range(1, min(n, CLIP_CAP) + 1))
        pick = rng.sample(range(n), count)       # NOTE (reference behaviour): these are clip NUMBERS used as frame ids
    elif method == "random_200":
        pick = rng.sample(range(n), min(n, 200))
    elif method == "uniform":
        step = min(subsample_factor, n)
        pick = range(0, n, step)[:min(n, CLIP_CAP)]
    else:
        raise ValueError("Clip sampling method %s not valid" % method)
    return np.array(pick, dtype=np.int64).reshape(-1)


def choose_videos(videos, required_shots, method, shot_cap, rng=_random):
    """reference :313-336"""
    have = len(videos)
    want = min(min(required_shots, shot_cap), have)
    if method == "specific":
        return videos[:want]
    if method == "fixed":
        return rng.sample(videos, want)
    if method == "random":
        return rng.sample(videos, rng.choice(range(1, min(have, shot_cap) + 1)))
    if method == "max":
        return rng.sample(videos, min(have, shot_cap))
    raise ValueError("Shot sampling method %s not valid" % method)


class ORBITDataset(torch.utils.data.Dataset):
    def __init__(self, root, way_method, object_cap, shot_methods, shots, video_types, subsample_factor, clip_methods,
                 clip_length, frame_size, frame_norm_method, annotations_to_load=(), filter_by_annotations=((), ()),
                 test_mode=False, with_cluster_labels=False, with_caps=False, logfile=None, frames="float", rng=None,
                 decode_pool=None):
        if annotations_to_load or any(filter_by_annotations) or with_cluster_labels:
            raise NotImplementedError("frame annotations / cluster labels are outside the recognition path (DESIGN.md §7)")
        if frames not in ("float", "uint8", "paths"):
            raise ValueError("frames must be 'float', 'uint8' or 'paths'")
        self.root, self.mode = root, os.path.basename(root)
        self.way_method, self.object_cap = way_method, object_cap
        self.shot_method = dict(zip(("context", "target"), shot_methods))
        self.shot = dict(zip(("context", "target"), shots))
        self.video_type = dict(zip(("context", "target"), video_types))
        self.clip_method = dict(zip(("context", "target"), clip_methods))
        self.subsample_factor, self.clip_length, self.frame_size = subsample_factor, int(clip_length), frame_size
        self.frame_norm_method = frame_norm_method
        self.normalize_stats = dict(zip(("mean", "std"), NORMALIZE_STATS[frame_norm_method]))
        self.test_mode, self.with_caps, self.logfile = test_mode, with_caps, logfile
        self.shot_cap = {"context": 15, "target": 15}
        self.frames, self.rng, self.decode_pool = frames, (rng if rng is not None else _random), decode_pool
        self.annotations_to_load, self.with_annotations = [], False
        self.users, self.user2objs, self.obj2user, self.obj2name, self.obj2vids = [], {}, {}, {}, {}
        self.video2id, self.vid2frames = {}, {}
        self._index()

    # ---- directory index -------------------------------------------------------------------------------------------------
    def _split_videos(self, obj_path):
        clean = sorted(os.listdir(os.path.join(obj_path, "clean")))
        if self.video_type["context"] == "clean" and self.video_type["target"] == "clean":
            k = min(5, len(clean) - 1)
            return {"context": clean[:k], "target": clean[k:]}
        if self.video_type["context"] == "clean" and self.video_type["target"] == "clutter":
            return {"context": clean, "target": sorted(os.listdir(os.path.join(obj_path, "clutter")))}
        return {"context": [], "target": []}

    def _index(self):
        next_obj = next_vid = 0
        for user in sorted(os.listdir(self.root)):
            mine = []
            user_path = os.path.join(self.root, user)
            for obj_name in sorted(os.listdir(user_path)):
                obj_path = os.path.join(user_path, obj_name)
                kept, frames_of = {"context": [], "target": []}, {}
                for which, names in self._split_videos(obj_path).items():
                    for name in names:
                        vp = os.path.join(obj_path, self.video_type[which], name)
                        jpgs = glob.glob(os.path.join(vp, "*.jpg"))
                        if len(jpgs) >= MIN_FRAMES[which]:
                            kept[which].append(vp)
                            frames_of[vp] = sorted(jpgs)
                if kept["context"] and kept["target"]:
                    mine.append(next_obj)
                    self.obj2user[next_obj], self.obj2name[next_obj], self.obj2vids[next_obj] = user, obj_name, kept
                    next_obj += 1
                    for vp in kept["context"] + kept["target"]:
                        self.video2id[vp], self.vid2frames[vp] = next_vid, frames_of[vp]
                        next_vid += 1
            if mine:
                self.users.append(user)
                self.user2objs[user] = mine
        self.num_users, self.num_objects = len(self.users), len(self.obj2name)

    def __len__(self):
        return self.num_users

    def get_user_objects(self, user):
        return self.user2objs[self.users[user]]

    # ---- sampling --------------------------------------------------------------------------------------------------------
    def compute_way(self, num_objects):
        top = min(num_objects, self.object_cap)
        if self.way_method == "random":
            return self.rng.choice(range(2, top + 1))
        if self.way_method == "max":
            return top
        raise ValueError("Way method %s not valid" % self.way_method)

    def sample_clips_from_a_video(self, frame_paths, sample_method):
        return clip_frame_indices(len(frame_paths), self.clip_length, sample_method, self.subsample_factor, self.rng)

    def _sample_set(self, videos, which):
        """-> clip paths [n][T] (object arrays), video id per clip, for the sampled videos of one object, in video order"""
        paths, vids = [], []
        for vp in videos:
            frame_paths = np.array(self.vid2frames[vp])
            rows = frame_paths[self.sample_clips_from_a_video(frame_paths, self.clip_method[which])]
            rows = rows.reshape(-1, self.clip_length)
            paths.extend(rows)
            vids.extend([self.video2id[vp]] * len(rows))
        return paths, vids

    def load_and_transform_frame(self, frame_path):
        """reference :422-431: to_tensor (HWC u8 -> CHW f32 / 255) then normalize ((x - mean) / std), fp32 throughout"""
        u8 = self._decode([frame_path])[0]
        return self._normalise(u8.unsqueeze(0))[0]

    def _decode(self, flat_paths):
        from .pipeline import decode_frames
        if len(flat_paths) == 0:
            return torch.empty(0, self.frame_size, self.frame_size, 3, dtype=torch.uint8)
        return torch.from_numpy(decode_frames(flat_paths, pool=self.decode_pool))

    def _normalise(self, u8_nhwc):
        x = u8_nhwc.permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
        mean = torch.tensor(self.normalize_stats["mean"], dtype=torch.float32).view(1, 3, 1, 1)
        std = torch.tensor(self.normalize_stats["std"], dtype=torch.float32).view(1, 3, 1, 1)
        return x.sub_(mean).div_(std)

    def load_clips(self, paths):
        """clip paths [n, T] -> clips in the configured representation (float: [n, T, 3, H, W] f32; uint8: [n, T, H, W, 3])"""
        paths = np.asarray(paths, dtype=object).reshape(-1, self.clip_length)
        if self.frames == "paths":
            return None
        u8 = self._decode(list(paths.reshape(-1)))
        if self.frames == "uint8":
            return u8.reshape(len(paths), self.clip_length, *u8.shape[1:])
        f = self._normalise(u8)
        return f.reshape(len(paths), self.clip_length, *f.shape[1:])

    def _finish_set(self, paths, labels, video_ids, by_video):
        paths = np.array(paths) if len(paths) else np.empty((0, self.clip_length), dtype=object)
        labels = torch.tensor(labels)
        if not by_video:
            order = list(range(len(paths)))
            self.rng.shuffle(order)       # the draws of the reference's random.shuffle over np.arange(n) (:517-518)
            order = np.array(order, dtype=np.int64)
            paths, labels = paths[order], labels[order]
            return self.load_clips(paths), paths, labels, {}
        video_ids = np.asarray(video_ids)
        frames_by_video, paths_by_video, labels_by_video, anns_by_video = [], [], [], []
        for vid in np.unique(video_ids):
            rows = video_ids == vid
            vp = paths[rows].reshape(-1)
            clips = self.load_clips(vp.reshape(-1, self.clip_length))
            frames_by_video.append(None if clips is None else clips.flatten(end_dim=1))
            paths_by_video.append(vp)
            labels_by_video.append(labels[torch.from_numpy(rows)][0])
            anns_by_video.append(None)
        return frames_by_video, paths_by_video, labels_by_video, anns_by_video

    def sample_task(self, task_objects, task_id=None):
        way = self.compute_way(len(task_objects))
        chosen = sorted(self.rng.sample(task_objects, way))
        label_of = {obj: i for i, obj in enumerate(chosen)}
        if self.with_caps:
            self.shot_cap = {"context": 5 if way >= 6 else 10, "target": 4 if way >= 6 else 8}
        sets = {w: {"paths": [], "labels": [], "vids": []} for w in ("context", "target")}
        names = []
        for obj in chosen:
            names.append(self.obj2name[obj])
            picked = {w: choose_videos(self.obj2vids[obj][w], self.shot[w], self.shot_method[w], self.shot_cap[w], self.rng)
                      for w in ("context", "target")}
            for w in ("context", "target"):
                p, v = self._sample_set(picked[w], w)
                sets[w]["paths"].extend(p)
                sets[w]["labels"].extend([label_of[obj]] * len(p))
                sets[w]["vids"].extend(v)
        c = self._finish_set(sets["context"]["paths"], sets["context"]["labels"], sets["context"]["vids"], False)
        t = self._finish_set(sets["target"]["paths"], sets["target"]["labels"], sets["target"]["vids"], self.test_mode)
        return {"context_clips": c[0], "context_paths": c[1], "context_labels": c[2], "context_annotations": c[3],
                "target_clips": t[0], "target_paths": t[1], "target_labels": t[2], "target_annotations": t[3],
                "object_list": names, "task_id": task_id}


class UserEpisodicORBITDataset(ORBITDataset):
    """One task per user: the user's own objects (reference :600-618)."""

    def __getitem__(self, index):
        user = self.users[index]
        return self.sample_task(self.user2objs[user], user)


class ObjectEpisodicORBITDataset(ORBITDataset):
    """Tasks drawn from all objects of all users (reference :620-637)."""

    def __getitem__(self, index):
        return self.sample_task(range(len(self.obj2vids)), None)
