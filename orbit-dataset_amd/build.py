"""Build liborbit_hip.so (gfx950) in-tree with hipcc.

Usage: python orbit-dataset_amd/build.py [--force] [--verbose]
Objects are compiled in parallel (one hipcc per .hip file) and linked into orbit-dataset_amd/lib/liborbit_hip.so.
No GPU is needed: hipcc cross-compiles for gfx950.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "liborbit_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["head.hip", "heads_extra.hip", "loss.hip", "conv_igemm.hip", "pw_rgemm.hip", "conv_bf3.hip", "ops.hip", "film.hip", "ingest.hip", "mbconv_rows.hip", "stem.hip", "extractor.hip", "train_ops.hip", "train_mbconv.hip", "conv_wgrad.hip", "extractor_train.hip",
           "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", INCLUDE, "-I", "/opt/rocm/include"]


def _runtime_libdir():
    """Directory holding the HIP runtime to bind to.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / librccl.so; stream handles and device pointers
    cross the C-ABI, so the library MUST use the same runtime instance as torch (two HIP runtimes in one
    process do not share streams). ORBIT_SYSTEM_ROCM=1 links the system ROCm instead (non-Python hosts).
    """
    if os.environ.get("ORBIT_SYSTEM_ROCM") == "1":
        return "/opt/rocm/lib"
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return "/opt/rocm/lib"


RTLIB = _runtime_libdir()
LINK = ["-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt", "-L" + RTLIB, "-lamdhip64", "-lrccl",
        "-Wl,-rpath," + RTLIB, "-Wl,--no-undefined"]


def _digest(paths):
    """Fingerprint of the sources and flags. Paths enter RELATIVE to the repository (and the flags with the checkout
    prefix removed), so the stamp that travels with a built .so matches on whatever path the tree is unpacked to."""
    root = os.path.dirname(HERE)
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.relpath(p, root).encode())
            h.update(f.read())
    h.update(" ".join(x.replace(root, "<repo>") for x in FLAGS + LINK).encode())
    return h.hexdigest()


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return _sources() + hdrs


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(OBJDIR, "stamp")
    digest = _digest(_deps())
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(_sources()))) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [HIPCC] + objs + LINK + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
