"""ctypes binding of liborbit_hip.so (the C-ABI declared in include/orbit_hip.h).

No torch types cross the boundary: tensors are handed over as raw device pointers (`Tensor.data_ptr()`)
plus sizes, and the current torch HIP stream as an opaque handle. There is NO CPU fallback: if the
library is missing or no GPU is visible, every entry point raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liborbit_hip.so")

_lib = None

P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "orbit_version": (c_int, []),
    "orbit_last_error": (c_char_p, []),
    "orbit_device_count": (c_int, []),
    "orbit_set_option": (c_int, [c_char_p, c_int]),
    "orbit_get_option": (c_int, [c_char_p]),
    "orbit_label_set": (c_int, [P, c_int, P, c_int, P, P]),
    "orbit_proto_configure": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "orbit_proto_finalize": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "orbit_proto_predict": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P, P, P]),
    "orbit_mean_pool": (c_int, [P, c_int, c_int, c_int, P, P]),
    "orbit_set_mean": (c_int, [P, c_int, c_int, P, P]),
    "orbit_history_mean_pool": (c_int, [P, c_int, c_int, c_int, P, P]),
    "orbit_extractor_create": (c_int, [c_char_p, c_int, c_int, POINTER(c_void_p)]),
    "orbit_extractor_create_ex": (c_int, [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "orbit_extractor_destroy": (None, [P]),
    "orbit_extractor_num_params": (c_int, [P]),
    "orbit_extractor_param_name": (c_char_p, [P, c_int]),
    "orbit_extractor_param_numel": (c_size_t, [P, c_int]),
    "orbit_extractor_load": (c_int, [P, c_char_p, P, c_size_t]),
    "orbit_extractor_load_async": (c_int, [P, c_char_p, P, c_size_t, P]),
    "orbit_extractor_load_all_async": (c_int, [P, P, c_int, P]),
    "orbit_extractor_finalize": (c_int, [P, P]),
    "orbit_extractor_output_size": (c_int, [P]),
    "orbit_extractor_film_slots": (c_int, [P]),
    "orbit_extractor_film_slot_channels": (c_int, [P, c_int]),
    "orbit_extractor_film_slot_name": (c_char_p, [P, c_int]),
    "orbit_extractor_film_size": (c_int, [P]),
    "orbit_extractor_workspace_bytes": (c_size_t, [P, c_int]),
    "orbit_extractor_macs_per_frame": (c_double, [P]),
    "orbit_extractor_forward": (c_int, [P, P, c_int, P, P, P, P, c_size_t, P]),
    "orbit_filmgen_create": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                     POINTER(c_void_p)]),
    "orbit_filmgen_destroy": (None, [P]),
    "orbit_filmgen_load": (c_int, [P, c_int, c_char_p, P, c_size_t]),
    "orbit_filmgen_load_all_async": (c_int, [P, P, c_int, P]),
    "orbit_filmgen_forward": (c_int, [P, P, P, P, P, P]),
    "orbit_op_conv2d": (c_int, [P, c_int, P, P, P, P, P, P] + [c_int] * 14 + [P]),
    "orbit_op_dwconv2d": (c_int, [P, P, P, P, P] + [c_int] * 11 + [P]),
    "orbit_op_conv2d_train": (c_int, [P, c_int, P, P, P] + [c_int] * 12 + [P, POINTER(c_int), P]),
    "orbit_op_dwconv2d_train": (c_int, [P, P, P, P, P] + [c_int] * 11 + [P, P]),
    "orbit_op_maxpool2d": (c_int, [P, P] + [c_int] * 9 + [P]),
    "orbit_op_avgpool": (c_int, [P, P, c_int, c_int, c_int, P]),
    "orbit_op_se_gate": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "orbit_op_mbconv_front": (c_int, [P] * 9 + [c_int] * 11 + [P]),
    "orbit_op_mbconv_front_partials": (c_int, [c_int] * 6),
    "orbit_op_stem_dw_front_partials": (c_int, [c_int] * 3),
    "orbit_op_stem_dw_front": (c_int, [P] * 9 + [c_int] * 12 + [P]),
    "orbit_dense_rows": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, P, P]),
    "orbit_spd_inverse": (c_int, [P, P, c_int, c_int, P]),
    "orbit_mahalanobis_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "orbit_mahalanobis_configure": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "orbit_mahalanobis_predict": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    "orbit_mahalanobis_predict_backward": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    "orbit_extractor_supports_training": (c_int, [P]),
    "orbit_extractor_tape_bytes": (c_size_t, [P, c_int]),
    "orbit_extractor_backward_workspace_bytes": (c_size_t, [P, c_int]),
    "orbit_extractor_grad_floats": (c_size_t, [P]),
    "orbit_extractor_param_offset": (c_size_t, [P, c_int]),
    "orbit_extractor_bn_stat_floats": (c_size_t, [P]),
    "orbit_extractor_export_bn_stats": (c_int, [P, P, P]),
    "orbit_extractor_train_forward": (c_int, [P, P, c_int, P, P, c_int, c_float, P, P, c_size_t, P]),
    "orbit_extractor_train_forward_ex": (c_int, [P, P, c_int, P, P, c_int, c_float, P, P, c_size_t, c_int, P]),
    "orbit_extractor_apply_deferred_bn_stats": (c_int, [P, P, c_size_t, c_int, c_float, P]),
    "orbit_extractor_backward": (c_int, [P, P, c_int, P, P, c_int, P, P, c_size_t, P, c_int, P, P, P, c_size_t, P]),
    "orbit_filmgen_grad_floats": (c_size_t, [P]),
    "orbit_filmgen_param_offset": (c_size_t, [P, c_int, c_char_p]),
    "orbit_filmgen_backward": (c_int, [P, P, P, P, P, P, P, P]),
    "orbit_proto_predict_backward": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, P, P]),
    "orbit_linear_head_backward": (c_int, [P, P, c_int, c_int, c_int, c_float, P, P, P]),
    "orbit_cross_entropy_forward": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P]),
    "orbit_cross_entropy_backward": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "orbit_op_bn_train_forward": (c_int, [P, c_int, c_int, P, P, c_float, c_float, P, P, P, c_int, P, P, P, P]),
    "orbit_op_bn_stats_from_gram": (c_int, [P, c_int, c_int, P, c_int, c_float, P, P, P]),
    "orbit_op_bn_backward": (c_int, [P, P, P, c_int, c_int, P, P, P, c_int, c_int, P, P, P, P, P]),
    "orbit_op_conv2d_dgrad": (c_int, [P, P, P, P] + [c_int] * 12 + [P]),
    "orbit_op_conv2d_wgrad": (c_int, [P, c_int, P, P] + [c_int] * 12 + [P]),
    "orbit_op_conv2d_wgrad_gated": (c_int, [P, P, P, P] + [c_int] * 5 + [P]),
    "orbit_op_maxpool2d_train": (c_int, [P, P, P] + [c_int] * 9 + [P]),
    "orbit_op_maxpool2d_backward": (c_int, [P, P, P] + [c_int] * 9 + [P]),
    "orbit_op_avgpool_backward": (c_int, [P, P, c_int, c_int, c_int, P]),
    "orbit_op_dwconv2d_backward": (c_int, [P, P, P, P, P] + [c_int] * 10 + [P]),
    "orbit_op_dwconv2d_dgrad_bn": (c_int, [P] * 7 + [c_int, P, P, P] + [c_int] * 10 + [P]),
    "orbit_op_dwconv2d_wgrad_xf": (c_int, [P, P, P, c_int, P, P] + [c_int] * 10 + [P]),
    "orbit_op_se_gate_backward": (c_int, [P] * 12 + [c_int] * 4 + [P]),
    "orbit_frames_from_uint8": (c_int, [P, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P]),
    "orbit_prof_enable": (c_int, [c_int]),
    "orbit_prof_collect": (c_int, [POINTER(c_double), POINTER(c_double), POINTER(ctypes.c_long)]),
    "orbit_prof_num_variants": (c_int, []),
    "orbit_prof_variant": (c_int, [c_int, ctypes.c_char_p, POINTER(ctypes.c_long), POINTER(c_double),
                                   POINTER(c_double), POINTER(c_double)]),
    "orbit_prof_set_roofs": (c_int, [c_double, c_double, c_double]),
    "orbit_prof_variant_floor": (c_int, [c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "orbit_runtime_init": (c_int, []),
    "orbit_comm_unique_id": (c_int, [P]),
    "orbit_comm_init": (c_int, [c_int, c_int, P]),
    "orbit_comm_world": (c_int, []),
    "orbit_comm_rank": (c_int, []),
    "orbit_allreduce_sum": (c_int, [P, c_size_t, P]),
    "orbit_extractor_train_graph_stats": (c_int, [P, P, P]),
    "orbit_comm_destroy": (None, []),
    "orbit_p2p_create": (c_int, [c_int, c_int, c_size_t, POINTER(c_void_p)]),
    "orbit_p2p_export": (c_int, [P, P]),
    "orbit_p2p_connect": (c_int, [P, P]),
    "orbit_p2p_allreduce_sum": (c_int, [P, P, c_size_t, P]),
    "orbit_p2p_allreduce_sum_sharded": (c_int, [P, P, c_size_t, P]),
    "orbit_p2p_error": (c_int, [P]),
    "orbit_p2p_memory_kind": (c_int, [P]),
    "orbit_p2p_destroy": (None, [P]),
}

EXPORTS = tuple(_SIGNATURES)


class OrbitHipError(RuntimeError):
    """Raised when the HIP library is unavailable or a call into it fails."""


def _hip_runtimes_mapped():
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return out


def load(path=None):
    """Load (once) and return the ctypes library; raises OrbitHipError if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("ORBIT_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise OrbitHipError(
            "liborbit_hip.so not found at %s — build it with `python __graft_entry__.py` "
            "(or `python orbit-dataset_amd/build.py`). There is no CPU fallback for this path." % path)
    # torch must own the HIP runtime instance: its bundled libamdhip64 (SONAME libamdhip64.so.7) is loaded
    # by `import torch`, and the dynamic loader then resolves our NEEDED entry to that same instance.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise OrbitHipError("cannot load %s: %s" % (path, e)) from e
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise OrbitHipError("two HIP runtimes are mapped into this process (%s): stream handles would not be "
                            "interchangeable. Rebuild the library against torch's runtime." % sorted(rts))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OrbitHipError("liborbit_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().orbit_last_error()
    return msg.decode() if msg else ""


def check(rc, what=""):
    if rc != 0:
        msg = last_error()
        if "not set - is the model personalised" in msg:
            raise AttributeError(msg)
        if rc == -1:
            raise ValueError("%s: %s" % (what or "liborbit_hip", msg))
        raise OrbitHipError("%s failed (code %d): %s" % (what or "liborbit_hip call", rc, msg))


_gpu_ok = False


def require_gpu():
    """Fail loudly unless a HIP device is usable through torch (checked once: is_available() costs ~0.4 ms)."""
    global _gpu_ok
    if _gpu_ok:
        return
    import torch
    load()
    if not torch.cuda.is_available():
        raise OrbitHipError("no HIP device is visible to torch; the ORBIT hot path has no CPU fallback")
    check(load().orbit_runtime_init(), "orbit_runtime_init")  # (current device; other devices: first launch on them)
    _gpu_ok = True


import threading

_tls = threading.local()  # the stream override is PER THREAD, as torch's current stream is: the staging thread of
#                           data/pipeline.TaskPrefetcher launches orbit_frames_from_uint8 on its copy stream while the main thread
#                           may sit inside a LITE `use_stream(side)` block - a process-global override sent that conversion kernel
#                           to the LITE side stream, unordered against the copy it reads


class use_stream:
    """Within the block the native entry points called FROM THIS THREAD are handed `stream` (a torch.cuda.Stream) instead of
    torch's current stream, while torch itself - its allocator included - stays on the current one: the caller orders the two
    streams with events (LITE's subset pass beside the cache pass, few_shot_recognisers._get_features_with_split_batch)."""

    def __init__(self, stream):
        self.handle = c_void_p(stream.cuda_stream)

    def __enter__(self):
        self.prev = getattr(_tls, "stream", None)
        _tls.stream = self.handle
        return self

    def __exit__(self, *exc):
        _tls.stream = self.prev
        return False


def stream_handle():
    """hipStream_t of torch's CURRENT stream on the current device, as an opaque pointer (or the stream a `use_stream` block
    of this thread names). Uses the raw accessor: torch.cuda.current_stream() re-checks device availability on every call
    (~80 us each)."""
    override = getattr(_tls, "stream", None)
    if override is not None:
        return override
    import torch
    try:
        return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:  # private accessors moved: fall back to the public API
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=None):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    import torch
    if not t.is_cuda:
        raise OrbitHipError("expected a device tensor, got a %s tensor" % t.device)
    if not t.is_contiguous():
        raise OrbitHipError("expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise OrbitHipError("expected dtype %s, got %s" % (dtype, t.dtype))
    return c_void_p(t.data_ptr())
