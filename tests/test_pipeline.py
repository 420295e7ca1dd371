"""Dataset-side input pipeline (orbit-dataset_amd/data/pipeline.py), host part: the ORBIT directory walk and task layout of
reference data/datasets.py:139-200,584-597 on a synthetic JPEG tree, and the decode step of :422-431 (PIL -> uint8, then
to_tensor + normalize - here restated as ((u8 / 255) - mean) / std, which the GPU kernel reproduces bit for bit)."""
import os

import numpy as np
import pytest
import torch

import orbit_dataset_amd  # noqa: F401
from orbit_dataset_amd.data import pipeline
from orbit_dataset_amd.data.utils import NORMALIZE_STATS


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("orbit"))
    n = pipeline.write_synthetic_orbit_directory(root, users=2, objects_per_user=3, clean_videos=3, clutter_videos=2,
                                                 frames_per_video=6, frame_size=32)
    return root, n


def test_directory_walk_follows_the_reference_layout(tree):
    root, n = tree
    assert n == 2 * 3 * 5 * 6
    d = pipeline.ORBITDirectory(root)  # context = clean, target = clutter (the reference's test default)
    assert d.users == ["P000", "P001"] and all(len(d.user2objs[u]) == 3 for u in d.users)
    for obj, vids in d.obj2vids.items():
        assert len(vids["context"]) == 3 and all("/clean/" in v for v in vids["context"])
        assert len(vids["target"]) == 2 and all("/clutter/" in v for v in vids["target"])
        for v in vids["context"] + vids["target"]:
            frames = d.vid2frames[v]
            assert frames == sorted(frames) and len(frames) == 6 and all(f.endswith(".jpg") for f in frames)
    # clean / clean: aim for 5 context videos, leave at least one target video (datasets.py:157-160)
    d2 = pipeline.ORBITDirectory(root, "clean", "clean")
    for vids in d2.obj2vids.values():
        assert len(vids["context"]) == 2 and len(vids["target"]) == 1


@pytest.mark.parametrize("T", [1, 3])
def test_user_task_layout(tree, T):
    root, _ = tree
    d = pipeline.ORBITDirectory(root)
    t = d.user_task("P001", clip_length=T)
    n_ctx = 3 * 3 * (6 // T)  # objects x clean videos x non-overlapping clips
    assert t["context_paths"].shape == (n_ctx, T) and t["context_labels"].tolist() == sorted(t["context_labels"].tolist())
    assert t["target_paths"].shape == (3 * 2 * 6, 1) and len(t["target_videos"]) == 6
    assert sorted(set(t["target_labels"].tolist())) == [0, 1, 2] and len(t["object_list"]) == 3
    for row in t["context_paths"]:  # a clip = T contiguous frames of ONE video
        assert len({os.path.dirname(p) for p in row}) == 1
        idx = [int(p[-9:-4]) for p in row]
        assert idx == list(range(idx[0], idx[0] + T))
    for (lo, hi), lab in zip(t["target_videos"], [0, 0, 1, 1, 2, 2]):
        assert hi - lo == 6 and set(t["target_labels"][lo:hi].tolist()) == {lab}


def test_decode_matches_pil_and_threads_agree(tree):
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    root, _ = tree
    d = pipeline.ORBITDirectory(root)
    paths = d.user_task("P000")["context_paths"].reshape(-1)[:10]
    a = pipeline.decode_frames(paths)
    with ThreadPoolExecutor(4) as pool:
        b = pipeline.decode_frames(paths, pool=pool)
    assert a.dtype == np.uint8 and a.shape == (10, 32, 32, 3) and np.array_equal(a, b)
    assert np.array_equal(a[3], np.asarray(Image.open(paths[3]).convert("RGB")))
    # the transform the GPU applies to these bytes = to_tensor + normalize of the reference (datasets.py:428-430)
    mean, std = NORMALIZE_STATS["imagenet"]
    x = torch.from_numpy(a[3]).permute(2, 0, 1).float().div(255.0)
    want = (x - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    assert want.shape == (3, 32, 32) and float(want.abs().max()) < 3.0


def test_directory_source_yields_reference_shaped_tasks(tree):
    root, _ = tree
    src = pipeline.DirectoryTaskSource(pipeline.ORBITDirectory(root), clip_length=2, workers=3)
    tasks = list(src)
    assert len(tasks) == 2
    t = tasks[0]
    assert t["context_clips"].dtype == torch.uint8 and t["context_clips"].shape == (27, 2, 32, 32, 3)
    assert t["target_clips"].shape == (36, 1, 32, 32, 3) and len(t["context_labels"]) == 27 and len(t["target_labels"]) == 36
