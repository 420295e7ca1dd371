"""TaskPrefetcher on hardware: tasks decoded from a JPEG tree, uploaded as 8-bit frames on a copy stream and normalised on
the GPU while the extractor works on the previous task, give logits BIT-EQUAL to the same frames held resident; the transform
is the reference's to_tensor + normalize (data/datasets.py:422-431)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd import synthetic  # noqa: E402
from orbit_dataset_amd.data import pipeline  # noqa: E402
from orbit_dataset_amd.data.utils import NORMALIZE_STATS, frames_from_uint8  # noqa: E402
from orbit_dataset_amd.model.few_shot_recognisers import SingleStepFewShotRecogniser  # noqa: E402


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("orbit"))
    pipeline.write_synthetic_orbit_directory(root, users=4, objects_per_user=3, clean_videos=2, clutter_videos=2,
                                             frames_per_video=5, frame_size=64)
    return root


def _model(device):
    model = SingleStepFewShotRecogniser("resnet18", False, "proto", 1, 8, False, 16, 1.0)
    synthetic.init_parameters_(model)
    model._set_device(device)
    model._send_to_device()
    model.set_test_mode(True)
    return model


def test_prefetched_tasks_give_bit_equal_logits(device, tree):
    model = _model(device)
    directory = pipeline.ORBITDirectory(tree)
    host_tasks = list(pipeline.DirectoryTaskSource(directory, workers=4))
    assert len(host_tasks) == 4
    want = []
    with torch.no_grad():
        for t in host_tasks:  # resident reference: the same bytes, normalised up front
            ctx = frames_from_uint8(t["context_clips"], device, channels_last=True)
            tgt = frames_from_uint8(t["target_clips"], device, channels_last=True)
            # the GPU transform is the reference transform, bit for bit
            mean, std = NORMALIZE_STATS["imagenet"]
            x = t["context_clips"][0, 0].permute(2, 0, 1).float().div(255.0)
            ref = (x - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
            assert torch.equal(ctx[0, 0].cpu(), ref)
            model.personalise(ctx, t["context_labels"].to(device))
            want.append(model.predict(tgt).clone())
            model._reset()
    got = []
    pf = pipeline.TaskPrefetcher(pipeline.DirectoryTaskSource(directory, workers=4), device, depth=2)
    with torch.no_grad():
        for t in pf:
            assert t["context_clips"].is_cuda and t["context_clips"].dtype == torch.float32
            assert t["context_clips"].shape[-3:] == (3, 64, 64) and t["context_labels"].is_cuda
            model.personalise(t["context_clips"], t["context_labels"])
            got.append(model.predict(t["target_clips"]).clone())  # (slot tensors are only valid until the next task)
            model._reset()
    pf.close()
    assert len(got) == len(want) == 4
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_prefetcher_with_pinned_host_tasks_and_more_tasks_than_slots(device):
    """Pre-decoded tasks (pinned, channels-first uint8 as the recogniser's own 8-bit path takes them): 7 tasks through 3
    slots, every one bit-equal to the resident run; a failing source surfaces in the consumer."""
    model = _model(device)
    g = torch.Generator().manual_seed(3)
    tasks = [{"context_clips": torch.randint(0, 256, (12, 1, 3, 64, 64), dtype=torch.uint8, generator=g).pin_memory(),
              "context_labels": torch.arange(12) % 3,
              "target_clips": torch.randint(0, 256, (9, 1, 3, 64, 64), dtype=torch.uint8, generator=g)} for _ in range(7)]
    with torch.no_grad():
        want = []
        for t in tasks:
            model.personalise(t["context_clips"], t["context_labels"].to(device))  # the recogniser's own uint8 upload path
            want.append(model.predict(t["target_clips"]).clone())
            model._reset()
        pf = pipeline.TaskPrefetcher(iter(tasks), device, depth=3)
        for i, t in enumerate(pf):
            model.personalise(t["context_clips"], t["context_labels"])
            assert torch.equal(model.predict(t["target_clips"]), want[i])
            model._reset()
        assert i == 6

        # float32 clips (the reference's task_dict layout): uploaded on the copy stream as they are
        tasks32 = [{"context_clips": torch.randn(12, 1, 3, 64, 64, generator=g), "context_labels": torch.arange(12) % 3,
                    "target_clips": torch.randn(9, 1, 3, 64, 64, generator=g).pin_memory()} for _ in range(4)]
        want32 = []
        for t in tasks32:
            model.personalise(t["context_clips"].to(device), t["context_labels"].to(device))
            want32.append(model.predict(t["target_clips"].to(device)).clone())
            model._reset()
        for i, t in enumerate(pipeline.TaskPrefetcher(iter(tasks32), device, depth=2)):
            assert t["context_clips"].is_cuda and t["context_clips"].dtype == torch.float32
            model.personalise(t["context_clips"], t["context_labels"])
            assert torch.equal(model.predict(t["target_clips"]), want32[i])
            model._reset()
        assert i == 3

    def broken():
        yield tasks[0]
        raise RuntimeError("decoder died")

    pf = pipeline.TaskPrefetcher(broken(), device)
    next(pf)
    with pytest.raises(RuntimeError, match="decoder died"):
        next(pf)


def test_learner_test_mode_over_a_jpeg_directory(device, tmp_path):
    """learner.py --mode test --data_root: the reference's per-user / per-video test loop (single-step-learner.py:298-375)
    fed by the reference-pinned task sampler (data/datasets.py, fixture G14) through the input pipeline; the per-task
    accuracies equal a plain loop over the same sampled tasks."""
    import random
    from orbit_dataset_amd import learner
    from orbit_dataset_amd.data.datasets import UserEpisodicORBITDataset
    tree = str(tmp_path / "test")
    pipeline.write_synthetic_orbit_directory(tree, users=3, objects_per_user=2, clean_videos=2, clutter_videos=1,
                                             frames_per_video=50, frame_size=64)   # 50 frames: the target-video floor
    args = learner.build_parser().parse_args(["--mode", "test", "--feature_extractor", "resnet18", "--frame_size", "64",
                                              "--data_root", tree, "--num_test_tasks", "3", "--num_workers", "3",
                                              "--subsample_factor", "5",
                                              "--batch_size", "16", "--results_path", str(tmp_path / "res.json")])
    L = learner.Learner(args)
    stats = L.run()["test"]
    assert stats["num_tasks"] == 3 and stats["target_frames"] == 3 * 2 * 1 * 50
    model = L.model
    ds = UserEpisodicORBITDataset(tree, "max", 15, ("max", "max"), (5, 2), ("clean", "clutter"), 5, ("uniform", "random_200"), 1,
                                  64, "imagenet", test_mode=True, frames="uint8", rng=random.Random(args.seed))
    accs = []
    with torch.no_grad():
        for t in pipeline.DatasetTaskSource(ds):
            assert t["context_clips"].shape[0] == 2 * 2 * 10  # objects x clean videos x every 5th of 50 frames
            ctx = frames_from_uint8(t["context_clips"], device, channels_last=True)
            tgt = frames_from_uint8(t["target_clips"], device, channels_last=True)[:, 0]
            model.personalise(ctx, t["context_labels"].to(device))
            per_video = []
            for lo, hi in t["target_videos"]:
                logits = model.predict_video(tgt[lo:hi])
                per_video.append((logits.argmax(1).cpu() == t["target_labels"][lo:hi]).float().mean().item())
            accs.append(sum(per_video) / len(per_video))
            model._reset()
    assert abs(stats["frame_acc"][0] - sum(accs) / len(accs)) < 1e-6


def test_prefetched_frames_equal_the_reference_dataset_tensors(device, tmp_path):
    """Fixture G14 (the REFERENCE's data/datasets.py on the stored JPEG tree): tasks sampled by data/datasets.py with the same
    seed, decoded to 8-bit, uploaded and normalised on the GPU by TaskPrefetcher, hold the frames the reference returned
    (to_tensor + normalize on the host, data/datasets.py:422-431) - compared bit for bit (the bar is 1 ulp)."""
    import os
    import random
    import numpy as np
    from test_datasets import CASES, GOLDEN, build, unpack_tree
    g14 = np.load(GOLDEN)
    root = unpack_tree(g14, str(tmp_path / "test"))
    for name in ("test_default", "train_cleanclean_T3"):
        ds = build(root, CASES[name], frames="uint8")
        random.seed(1991 + len(name))
        pf = pipeline.TaskPrefetcher(pipeline.DatasetTaskSource(ds, [0]), device, depth=2)
        t = next(pf)
        key = name + "_r0_i0"
        want = torch.from_numpy(g14[key + "_context_clips"])
        got = t["context_clips"].cpu()
        assert got.shape == want.shape and got.dtype == torch.float32
        ulp = (got.view(torch.int32) - want.view(torch.int32)).abs().max().item()
        assert ulp <= 1, "%s: %d ulp" % (name, ulp)
        assert torch.equal(t["context_labels"].cpu(), torch.from_numpy(g14[key + "_context_labels"]))
        if CASES[name]["test_mode"]:
            lo, hi = t["target_videos"][0]
            want_t = torch.from_numpy(g14[key + "_target0_frames"])
            got_t = t["target_clips"][lo:hi, 0].cpu()
            assert (got_t.view(torch.int32) - want_t.view(torch.int32)).abs().max().item() <= 1
            assert set(t["target_labels"][lo:hi].tolist()) == {int(g14[key + "_target0_label"])}
        pf.close()
