"""orbit-dataset_amd: MI355X (gfx950) native implementation of ORBIT's episodic few-shot recognition hot path.

The directory name carries a hyphen (the build contract's name); import it through the repo-root alias
module `orbit_dataset_amd`. Layout:
  csrc/      hand-written HIP kernels + the C-ABI (include/orbit_hip.h) -> lib/liborbit_hip.so
  _lib.py    ctypes binding (raw pointers + stream handle; no torch types cross the boundary)
  model/     host-side mirror of the reference's model/ interface for this path
  data/      the batch/clip helpers the path's callers use
  synthetic.py  ORBIT-shaped synthetic tasks + deterministic parameters (no dataset / checkpoints offline)
"""
__version__ = "0.1.0"
