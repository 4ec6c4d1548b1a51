"""ctypes binding of ``libhan_hip.so`` (the C ABI declared in ``include/han_hip.h``).

The library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``).
There is NO fallback: if the shared object is missing or a symbol is absent the
import of anything that computes raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhan_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ("node_attn.hip", "project.hip", "sem_attn.hip", "loss_opt.hip")
# node_attn.hip: the dense K2 kernels (node_attn_dense.h) keep 8-16 fp32 MFMA accumulators in a rolled loop; with the
# default AGPR form hipcc shuffles them through v_accvgpr_read / _mov / _write every iteration (each a wait for the
# matrix pipe); the VGPR form of the MFMA destination has no such traffic
EXTRA_FLAGS = {"node_attn.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form")}

P = c_void_p
I64 = c_int64


class HanRowSplit(ctypes.Structure):
    """han_row_split_t of include/han_hip.h."""
    _fields_ = [("split_deg", c_int64), ("n_long", c_int64), ("n_chunks", c_int64),
                ("long_rows", c_void_p), ("long_ptr", c_void_p), ("chunk_long", c_void_p),
                ("chunk_start", c_void_p), ("chunk_end", c_void_p), ("workspace", c_void_p),
                ("workspace_bytes", c_size_t), ("n_short", c_int64), ("n_mid", c_int64),
                ("short_rows", c_void_p), ("mid_rows", c_void_p)]


class HanDense(ctypes.Structure):
    """han_dense_t of include/han_hip.h."""
    _fields_ = [("bits", c_void_p), ("ld_words", c_int64), ("n_table", c_int64), ("workspace", c_void_p),
                ("workspace_bytes", c_size_t)]


# name -> (restype, argtypes); mirrors include/han_hip.h one to one
SIGNATURES = {
    "han_abi_version": (c_int, []),
    "han_error_string": (c_char_p, [c_int]),
    "han_project_fwd_workspace": (c_size_t, [I64, c_int, c_int, c_int]),
    "han_project_fwd_multi_workspace": (c_size_t, [I64, c_int, c_int, c_int, c_int]),
    "han_project_keep_bytes": (c_size_t, [I64, c_int, I64, c_int, c_int]),
    "han_project_fwd": (c_int, [P, c_int, I64, P, P, P, P, P, P, c_int, P, P, P, c_size_t, I64, c_int, c_int, c_int,
                                c_float, c_float, c_uint64, P, I64, P, c_int, P]),
    "han_project_fwd_multi": (c_int, [P, c_int, I64, P, P, P, P, P, P, c_int, P, P, P, c_size_t, I64, c_int, c_int, c_int,
                                      c_int, c_float, c_float, P, P, I64, P, c_int, P]),
    "han_project_bwd_workspace": (c_size_t, [I64, c_int, c_int, c_int]),
    "han_project_bwd": (c_int, [P, c_int, I64, P, P, P, c_size_t, I64, c_int, c_int, c_int, c_float,
                                c_uint64, P, I64, P, P]),
    "han_project_bwd_input": (c_int, [P, P, P, I64, I64, c_int, c_int, c_int, c_float, c_uint64, P, I64, P]),
    "han_row_split_workspace": (c_size_t, [I64]),
    "han_node_attn_fwd": (c_int, [P, P, P, P, c_int, P, P, P, P, P, P, P, P, I64, P, P, P, P, I64, I64, c_int,
                                  c_int, c_float, c_float, c_float, c_uint64, P, I64, c_int, c_int, P, P, P]),
    "han_node_attn_dense_workspace": (c_size_t, [I64, I64, c_int]),
    "han_csr_to_bitmask": (c_int, [P, P, I64, I64, P, I64, P, P]),
    "han_node_attn_coefs": (c_int, [P, P, P, P, P, P, P, c_int, I64, I64, c_int, c_int, c_float, c_float,
                                    c_uint64, P, I64, P]),
    "han_node_attn_bwd_workspace": (c_size_t, [I64, c_int, c_int]),
    "han_gs_row_bytes": (c_size_t, [c_int, c_int, c_int]),
    "han_node_attn_bwd_rows": (c_int, [P, I64, P, I64, P, P, P, P, P, P, P, c_int, P, P, P, c_size_t, I64,
                                       c_int, c_int, c_int, P]),
    "han_node_attn_bwd_cols": (c_int, [P, P, P, P, P, P, c_int, P, P, P, P, P, P, I64, I64, c_int, c_int,
                                       c_float, c_float, c_float, c_uint64, P, I64, I64, c_int, P, P, P]),
    "han_score_param_bwd_workspace": (c_size_t, [I64, c_int, c_int]),
    "han_score_param_bwd": (c_int, [P, c_int, P, P, P, P, P, P, P, c_size_t, I64, c_int, c_int, P]),
    "han_sem_attn_fwd": (c_int, [P, P, P, P, P, P, I64, c_int, c_int, c_int, c_int, P]),
    "han_sem_attn_bwd_workspace": (c_size_t, [I64, c_int, c_int, c_int]),
    "han_sem_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_size_t, I64, c_int, c_int,
                                 c_int, c_int, P]),
    "han_classifier_workspace": (c_size_t, [I64, c_int, c_int, c_int]),
    "han_classifier_loss": (c_int, [P, P, P, P, P, c_float, P, P, P, P, P, P, c_size_t, I64,
                                    c_int, c_int, c_int, P]),
    "han_classifier_bwd_workspace": (c_size_t, [I64, c_int, c_int, c_int]),
    "han_classifier_bwd": (c_int, [P, P, P, P, P, P, P, P, c_size_t, I64, c_int, c_int, c_int, P]),
    "han_adam_step": (c_int, [P, P, P, P, I64, c_float, c_float, c_float, c_float, c_float, P, P]),
    "han_l2_half_sumsq": (c_int, [P, I64, P, P, c_size_t, P]),
    "han_bias_row_counts": (c_int, [P, I64, I64, P, P]),
    "han_bias_fill_csr": (c_int, [P, I64, I64, P, P, P]),
}

ABI_VERSION = 6
_lib = None


class HanLibraryError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``han_amd/libhan_hip.so``: one object per source
    (compiled in parallel, re-compiled only when the source or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(_HERE, "..", "include", "han_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + list(EXTRA_FLAGS.get(s, ())) +
                        ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise HanLibraryError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs):
        tmp = f"{LIB_PATH}.{os.getpid()}.tmp"      # link aside, then rename: a waiting process never sees a partial file
        run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs)
        os.replace(tmp, LIB_PATH)
    return LIB_PATH


def load():
    """Load the library and bind every symbol of include/han_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HanLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "han_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HanLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.han_abi_version()
    if v != ABI_VERSION:
        raise HanLibraryError(f"libhan_hip.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


CALLS = 0      # C-ABI entry-point calls so far in this process (bench.py: calls recorded into a captured epoch)


def check(code: int, what: str = ""):
    global CALLS
    CALLS += 1
    if code != 0:
        msg = load().han_error_string(code)
        raise HanLibraryError(f"{what or 'libhan_hip'} failed ({code}): "
                              f"{msg.decode() if msg else '?'}")
