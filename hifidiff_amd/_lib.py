"""ctypes binding of libhifidiff_hip.so (C-ABI: include/hifidiff_hip.h) and its in-tree build.

There is no CPU fallback: if the shared library is missing or no gfx950 device is present, every
entry point raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` (hipcc
cross-compiles for gfx950 without a GPU).
"""
import ctypes
import os
import subprocess

# PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be loaded BEFORE this
# library so that our DT_NEEDED libamdhip64.so.7 resolves to that same runtime instance: one HIP runtime
# per process, shared streams and allocations.  Loading ours first pulls in /opt/rocm's copy and the
# process ends up with two runtimes (torch then reports "No HIP GPUs are available").
import torch  # noqa: F401  (load order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhifidiff_hip.so")
# translation units (compiled in parallel, one hipcc each) and the headers they include
UNITS = [os.path.join(_HERE, "csrc", f) for f in ("hd_lib.hip", "hd_aux.hip", "hd_stages.hip", "hd_strip.hip", "hd_dispatch_ln.hip", "hd_dispatch_lnface.hip", "hd_dispatch_bf16.hip", "hd_dispatch_misc.hip")]
SOURCES = UNITS + [os.path.join(_HERE, "csrc", f) for f in ("hd_gemm.hpp", "hd_dispatch.hpp", "hd_kernels.hpp", "hd_chain.hpp", "hd_conv.hpp", "hd_cr.hpp", "hd_vae.hpp", "hd_internal.hpp", "hd_stage_api.hpp", "hd_xcd.hpp", "hd_xcd2.hpp", "hd_face.hpp", "hd_strip.hpp", "hd_wide.hpp", "hd_end.hpp")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hifidiff_hip.h")


def unit_deps(unit):
    """The unit and every file it includes with #include "..." (recursively): what its object file depends on."""
    import re
    seen, todo = [], [unit]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        with open(f) as fh:
            for inc in re.findall(r'^\s*#include\s+"([^"]+)"', fh.read(), flags=re.M):
                todo.append(os.path.join(os.path.dirname(f), inc))
    return seen

EXPORTS = [
    "hd_create", "hd_create_unconditional", "hd_prepare_unconditional", "hd_cr_create", "hd_cr_forward", "hd_vae_create", "hd_vae_encode", "hd_vae_decode", "hd_destroy", "hd_last_error", "hd_load_weights", "hd_finalize_weights", "hd_prepare",
    "hd_prepare_from_priors", "hd_fpg", "hd_idc", "hd_eps", "hd_sample", "hd_scheduler_step", "hd_num_ops", "hd_num_chains",
    "hd_debug_limit_ops", "hd_debug_op_name", "hd_debug_read_op", "hd_debug_read", "hd_debug_write", "hd_set_option", "hd_get_option", "hd_check",
    "hd_set_profiling", "hd_get_profile",
]


class TensorDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int64 * 4), ("is_device", ctypes.c_int32)]


class Schedule(ctypes.Structure):
    _fields_ = [("n_steps", ctypes.c_int32), ("timesteps", ctypes.POINTER(ctypes.c_float)),
                ("coef", ctypes.POINTER(ctypes.c_float))]


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared: compiles every HIP kernel of the path into the in-tree .so."""
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in SOURCES + [HEADER])
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-function"]
    objdir = os.path.join(_HERE, "csrc", "build")
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.splitext(os.path.basename(u))[0] + ".o") for u in UNITS]
    # a unit is recompiled when it or a header it includes is newer than its object file
    stale = [(u, o) for u, o in zip(UNITS, objs)
             if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(d) for d in unit_deps(u))]
    cmds = [[hipcc] + flags + ["-c", u, "-o", o] for u, o in stale]
    if verbose:
        for c in cmds:
            print(" ".join(c))
    procs = [subprocess.Popen(c) for c in cmds]                 # the kernel instantiations dominate: one process per unit
    failed = [u for (u, _), pr in zip(stale, procs) if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(os.path.basename(f) for f in failed))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is not built (no CPU fallback exists); run __graft_entry__.build()")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
    L.hd_create.argtypes = [ctypes.POINTER(vp), i32, i32]
    L.hd_create_unconditional.argtypes = [ctypes.POINTER(vp), i32, i32]
    L.hd_prepare_unconditional.argtypes = [vp, i32, vp]
    L.hd_cr_create.argtypes = [ctypes.POINTER(vp), i32]
    L.hd_cr_forward.argtypes = [vp, i32, vp, vp, vp]
    L.hd_vae_create.argtypes = [ctypes.POINTER(vp), i32]
    L.hd_vae_encode.argtypes = [vp, i32, i32, i32, vp, i32, vp, u64, vp, vp, vp]
    L.hd_vae_decode.argtypes = [vp, i32, i32, vp, vp, vp]
    L.hd_destroy.argtypes = [vp]; L.hd_destroy.restype = None
    L.hd_last_error.argtypes = [vp]; L.hd_last_error.restype = ctypes.c_char_p
    L.hd_load_weights.argtypes = [vp, ctypes.POINTER(TensorDesc), i32]
    L.hd_finalize_weights.argtypes = [vp]
    L.hd_prepare.argtypes = [vp, i32, vp, vp, vp, vp]
    L.hd_prepare_from_priors.argtypes = [vp, i32, ctypes.POINTER(vp), vp, vp]
    L.hd_fpg.argtypes = [vp, i32, vp, ctypes.POINTER(vp), vp]
    L.hd_idc.argtypes = [vp, i32, vp, vp, vp]
    L.hd_eps.argtypes = [vp, vp, vp, i32, vp, vp]
    L.hd_sample.argtypes = [vp, vp, ctypes.POINTER(Schedule), vp, u64, vp]
    L.hd_scheduler_step.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float), vp, u64, i32, i64, vp]
    L.hd_num_ops.argtypes = [vp, i32]
    L.hd_num_chains.argtypes = [vp]
    L.hd_debug_limit_ops.argtypes = [vp, i32, i32]
    L.hd_debug_op_name.argtypes = [vp, i32, i32]; L.hd_debug_op_name.restype = ctypes.c_char_p
    L.hd_debug_read_op.argtypes = [vp, i32, i32, vp, i64]; L.hd_debug_read_op.restype = i64
    L.hd_debug_read.argtypes = [vp, ctypes.c_char_p, vp, i64]; L.hd_debug_read.restype = i64
    L.hd_debug_write.argtypes = [vp, ctypes.c_char_p, vp, i64]
    L.hd_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    L.hd_get_option.argtypes = [vp, ctypes.c_char_p]
    L.hd_check.argtypes = [vp]
    L.hd_set_profiling.argtypes = [vp, i32]
    L.hd_get_profile.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("hd_num_ops",):
            fn.restype = ctypes.c_int
    _lib = L
    return L


class HipError(RuntimeError):
    pass


def check(rc, ctx=None):
    """Negative return -> RuntimeError(hd_last_error()), like the reference raising on bad shapes."""
    if rc is not None and rc < 0:
        msg = lib().hd_last_error(ctx)
        raise HipError(f"hifidiff_hip error {rc}: {msg.decode() if msg else '?'}")
    return rc
