#!/usr/bin/env python
"""Entry point with the reference's name and flags (reference single-step-learner.py:48-51), driving the
MI355X-native recogniser on synthetic ORBIT-shaped tasks. See orbit-dataset_amd/learner.py.

    python single-step-learner.py --mode test --feature_extractor efficientnet_b0 --classifier proto \
        --frame_size 224 --batch_size 256 --num_test_tasks 8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 single-step-learner.py --mode test ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd.learner import main  # noqa: E402

if __name__ == "__main__":
    main()
