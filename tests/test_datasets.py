"""orbit-dataset_amd/data/datasets.py against fixture G14: what the REFERENCE's data/datasets.py returned (recorded by
tests/golden/make_golden.py:g14_datasets, which imports the reference's module) on the JPEG tree stored in the fixture, for
the same `random` seeds - index (users, objects, video ids, the 50-frame target floor), way / video / clip sampling
(max, random, random_200, uniform, last-frame padding at clip_length 3 / 4), train-mode shuffling, test-mode grouping by video,
and the decoded + normalised frames (reference data/datasets.py:104-205,289-336,422-522,540-598)."""
import os
import random

import numpy as np
import pytest
import torch

import orbit_dataset_amd  # noqa: F401
from orbit_dataset_amd.data import datasets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "G14_datasets.npz")

CASES = {  # the constructor arguments make_golden.py used (kept in step by test_case_list_matches_fixture)
    "test_default": dict(way_method="max", object_cap=15, shot_methods=("max", "max"), shots=(5, 2), video_types=("clean", "clutter"),
                         subsample_factor=3, clip_methods=("uniform", "random_200"), clip_length=1, test_mode=True, with_caps=False),
    "test_T4_max": dict(way_method="max", object_cap=2, shot_methods=("specific", "fixed"), shots=(2, 1), video_types=("clean", "clutter"),
                        subsample_factor=2, clip_methods=("max", "max"), clip_length=4, test_mode=True, with_caps=False),
    "train_random": dict(way_method="random", object_cap=15, shot_methods=("random", "random"), shots=(5, 2),
                         video_types=("clean", "clutter"), subsample_factor=4, clip_methods=("uniform", "random"), clip_length=1,
                         test_mode=False, with_caps=True),
    "train_cleanclean_T3": dict(way_method="max", object_cap=15, shot_methods=("fixed", "max"), shots=(3, 2),
                                video_types=("clean", "clean"), subsample_factor=1, clip_methods=("max", "max"), clip_length=3,
                                test_mode=False, with_caps=False),
    "train_r200_uniform": dict(way_method="random", object_cap=2, shot_methods=("max", "fixed"), shots=(5, 1),
                               video_types=("clean", "clutter"), subsample_factor=5, clip_methods=("random_200", "uniform"),
                               clip_length=1, test_mode=False, with_caps=False),
}


@pytest.fixture(scope="module")
def g14():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def tree(g14, tmp_path_factory):
    return unpack_tree(g14, str(tmp_path_factory.mktemp("g14") / "test"))


def unpack_tree(g14, root):
    files, blob, off = g14["tree_files"], g14["tree_blob"], g14["tree_offsets"]
    for i, rel in enumerate(files):
        path = os.path.join(root, str(rel))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(blob[off[i]:off[i + 1]].tobytes())
    return root


def build(root, kw, frames="float"):
    return datasets.UserEpisodicORBITDataset(root, kw["way_method"], kw["object_cap"], kw["shot_methods"], kw["shots"],
                                             kw["video_types"], kw["subsample_factor"], kw["clip_methods"], kw["clip_length"], 16,
                                             "imagenet", [], ([], []), kw["test_mode"], False, kw["with_caps"], None, frames=frames)


def file_numbers(g14, root, paths):
    no = {str(f): i for i, f in enumerate(g14["tree_files"])}
    a = np.asarray(paths)
    return np.array([no[os.path.relpath(p, root)] for p in a.reshape(-1)], dtype=np.int32).reshape(a.shape)


def test_case_list_matches_fixture(g14):
    assert sorted(CASES) == sorted(str(c) for c in g14["case_names"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_index_matches_reference(g14, tree, name):
    ds = build(tree, CASES[name], frames="paths")
    assert ds.users == [str(u) for u in g14[name + "_users"]]
    assert ds.num_objects == int(g14[name + "_num_objects"]) and len(ds) == len(ds.users)
    by_id = sorted(ds.video2id.items(), key=lambda kv: kv[1])
    assert [os.path.relpath(p, tree) for p, _ in by_id] == [str(v) for v in g14[name + "_video_ids"]]
    assert [len(ds.vid2frames[p]) for p, _ in by_id] == g14[name + "_video_frames"].tolist()
    for frames in ds.vid2frames.values():
        assert frames == sorted(frames)


@pytest.mark.parametrize("name", sorted(CASES))
def test_sampled_tasks_match_reference(g14, tree, name):
    """Same seed, same order of __getitem__ calls -> the same tasks: paths, labels, frames."""
    kw = CASES[name]
    ds = build(tree, kw)
    random.seed(1991 + len(name))
    for rep in range(2):
        for idx in range(len(ds)):
            t = ds[idx]
            key = "%s_r%d_i%d" % (name, rep, idx)
            assert t["object_list"] == [str(o) for o in g14[key + "_objects"]]
            assert t["task_id"] == ds.users[idx]
            np.testing.assert_array_equal(file_numbers(g14, tree, t["context_paths"]), g14[key + "_context_paths"])
            assert t["context_labels"].dtype == torch.int64
            np.testing.assert_array_equal(t["context_labels"].numpy(), g14[key + "_context_labels"])
            n = len(t["context_paths"])
            assert t["context_clips"].shape == (n, kw["clip_length"], 3, 16, 16) and t["context_clips"].dtype == torch.float32
            np.testing.assert_array_equal(t["context_clips"].double().sum(dim=(1, 2, 3, 4)).numpy(), g14[key + "_context_clips_sum"])
            if key + "_context_clips" in g14.files:
                np.testing.assert_array_equal(t["context_clips"].numpy(), g14[key + "_context_clips"])  # bit-exact frames
            if kw["test_mode"]:
                assert len(t["target_paths"]) == int(g14[key + "_target_videos"]) == len(t["target_clips"]) == len(t["target_labels"])
                for v, (fr, pa, la) in enumerate(zip(t["target_clips"], t["target_paths"], t["target_labels"])):
                    np.testing.assert_array_equal(file_numbers(g14, tree, pa), g14[key + "_target%d_paths" % v])
                    assert la.dim() == 0 and int(la) == int(g14[key + "_target%d_label" % v])
                    assert fr.shape == (len(pa), 3, 16, 16)
                    np.testing.assert_array_equal(fr.double().sum(dim=(1, 2, 3)).numpy(), g14[key + "_target%d_frames_sum" % v])
                    if key + "_target%d_frames" % v in g14.files:
                        np.testing.assert_array_equal(fr.numpy(), g14[key + "_target%d_frames" % v])
            else:
                np.testing.assert_array_equal(file_numbers(g14, tree, t["target_paths"]), g14[key + "_target_paths"])
                np.testing.assert_array_equal(t["target_labels"].numpy(), g14[key + "_target_labels"])
                np.testing.assert_array_equal(t["target_clips"].double().sum(dim=(1, 2, 3, 4)).numpy(), g14[key + "_target_clips_sum"])


def test_private_rng_and_uint8_frames_agree_with_float(g14, tree):
    """rng=: an own random.Random gives the same tasks as the seeded module; frames='uint8' holds the decoded bytes whose
    to_tensor + normalize is the float form (what TaskPrefetcher reproduces on the GPU)."""
    kw = CASES["train_random"]
    a, b = build(tree, kw), build(tree, kw, frames="uint8")
    random.seed(7)
    b.rng = random.Random(7)
    ta, tb = a[1], b[1]
    np.testing.assert_array_equal(ta["context_paths"], tb["context_paths"])
    assert tb["context_clips"].dtype == torch.uint8 and tb["context_clips"].shape[-1] == 3
    x = tb["context_clips"].flatten(end_dim=1)
    np.testing.assert_array_equal(a._normalise(x).reshape(ta["context_clips"].shape).numpy(), ta["context_clips"].numpy())


def test_clip_sampling_edges():
    f = datasets.clip_frame_indices
    assert f(5, 4, "max").tolist() == [0, 1, 2, 3, 4, 4, 4, 4]                 # padded with the last frame (:443-446)
    assert f(2500, 1, "max").tolist() == list(range(1000))                      # frame cap (:441)
    assert f(100, 1, "uniform", subsample_factor=30).tolist() == [0, 30, 60, 90]
    assert f(3, 1, "uniform", subsample_factor=30).tolist() == [0]              # factor clipped to the clip count (:464)
    assert len(f(1000, 1, "uniform", subsample_factor=1)) == 200                # clip cap (:463)
    assert sorted(f(7, 1, "random_200", rng=random.Random(0)).tolist()) == list(range(7))
    with pytest.raises(ValueError):
        f(4, 1, "nope")
    with pytest.raises(NotImplementedError):
        datasets.ORBITDataset(".", "max", 15, ("max", "max"), (5, 2), ("clean", "clutter"), 30, ("max", "max"), 1, 16, "imagenet",
                              annotations_to_load=["object_bounding_box"])


def test_dataset_task_source_layout(g14, tree):
    """data/pipeline.DatasetTaskSource: the host-task layout TaskPrefetcher uploads - a test-mode target set (a list of videos)
    becomes one uint8 tensor [M, 1, H, W, 3] + row ranges + one label per frame; a train-mode target set passes through."""
    from orbit_dataset_amd.data import pipeline
    ds = build(tree, CASES["test_default"], frames="uint8")
    random.seed(5)
    want = ds[1]
    random.seed(5)
    (task,) = list(pipeline.DatasetTaskSource(ds, [1]))
    assert task["task_id"] == ds.users[1] and task["object_list"] == want["object_list"]
    assert task["context_clips"].dtype == torch.uint8 and torch.equal(task["context_clips"], want["context_clips"])
    lens = [len(v) for v in want["target_clips"]]
    assert task["target_videos"] == [(sum(lens[:i]), sum(lens[:i + 1])) for i in range(len(lens))]
    assert task["target_clips"].shape == (sum(lens), 1, 16, 16, 3)
    for (lo, hi), frames, lab in zip(task["target_videos"], want["target_clips"], want["target_labels"]):
        assert torch.equal(task["target_clips"][lo:hi, 0], frames)
        assert task["target_labels"][lo:hi].tolist() == [int(lab)] * (hi - lo)
    train = build(tree, CASES["train_random"], frames="uint8")
    random.seed(6)
    (t2,) = list(pipeline.DatasetTaskSource(train, [0]))
    assert t2["target_clips"].dim() == 5 and "target_videos" not in t2 and len(t2["target_labels"]) == len(t2["target_clips"])
    with pytest.raises(ValueError):
        pipeline.DatasetTaskSource(build(tree, CASES["train_random"], frames="float"))
