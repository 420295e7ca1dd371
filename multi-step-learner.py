#!/usr/bin/env python
"""Entry point with the reference's name and flags (reference multi-step-learner.py:44-47): the FineTuner baseline on the
MI355X-native recogniser and backward kernels, synthetic ORBIT-shaped tasks. See orbit-dataset_amd/learner.py.

    python multi-step-learner.py --feature_extractor resnet18 --frame_size 84 --adapt_features \
        --personalize_num_grad_steps 50 --personalize_learning_rate 0.001 --num_test_tasks 4
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import orbit_dataset_amd  # noqa: E402,F401
from orbit_dataset_amd.learner import main_multistep  # noqa: E402

if __name__ == "__main__":
    main_multistep()
