"""ctypes binding of librflu.so (include/rflu.h).  The library is the product; this file only marshals pointers.

There is no CPU fallback anywhere in this package: if the shared library is missing or no gfx950 device is visible the
calls raise ``RfluError``.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFLU_LIB") or os.path.join(HERE, "librflu.so")   # RFLU_LIB: experiment builds (scripts/)

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_p = ctypes.c_void_p
c_u64 = ctypes.c_uint64
c_dbl = ctypes.c_double


class RfluError(RuntimeError):
    """A runtime (HIP / argument) failure reported by librflu -- never a numerical condition."""


# every exported symbol of include/rflu.h: name -> (restype, argtypes)
_TYPED = {
    "rflu_getrf_{s}": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_i64, c_p]),
    "rflu_getrf_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_i64, c_p]),
    "rflu_getrf_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_i64, c_p]),
    "rflu_getrs_{s}": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_i64]),
    "rflu_getrs_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_i64]),
    "rflu_getrs_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_p, c_i64]),
    "rflu_panel_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_int, c_p]),
    "rflu_laswp_rm_{s}_dev": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64]),
    "rflu_trsm_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_i64]),
    "rflu_gemm_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64]),
    "rflu_cm_to_rm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_i64]),
    "rflu_rm_to_cm_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_i64]),
    "rflu_butterfly_mul_{s}_dev": (c_int, [c_p, c_i64, c_p, c_i64, c_p]),
    "rflu_butterfly_vec_{s}_dev": (c_int, [c_p, c_i64, c_i64, c_p, c_i64, c_p, c_int]),
    "rflu_fill_uniform_{s}_dev": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_int, c_u64, c_i64, c_i64, c_i64, c_dbl]),
}
_PLAIN = {
    "rflu_create": (c_int, [ctypes.POINTER(c_p), c_int]),
    "rflu_destroy": (c_int, [c_p]),
    "rflu_reload_tuning": (c_int, [c_p]),
    "rflu_last_error": (ctypes.c_char_p, []),
    "rflu_version": (c_int, []),
    "rflu_set_stream": (c_int, [c_p, c_p]),
    "rflu_synchronize": (c_int, [c_p]),
    "rflu_last_path": (c_int, [c_p]),
    "rflu_update_stream": (c_int, [c_p, ctypes.POINTER(c_p)]),
    "rflu_debug_heat": (c_int, [c_p, ctypes.c_double]),
    "rflu_debug_gate_stamps": (c_int, [c_p, c_p]),
    "rflu_debug_engine_acct": (c_int, [c_p, c_p]),
    "rflu_mgpu_create": (c_int, [ctypes.POINTER(c_p), c_int, ctypes.POINTER(c_int)]),
    "rflu_mgpu_reload_tuning": (c_int, [c_p]),
    "rflu_mgpu_destroy": (c_int, [c_p]),
    "rflu_mgpu_ndev": (c_int, [c_p]),
    "rflu_mgpu_is_fake": (c_int, [c_p]),
    "rflu_mgpu_collectives": (c_i64, [c_p]),
    "rflu_mgpu_local_cols": (c_i64, [c_i64, c_i64, c_int, c_i64, c_int]),
    "rflu_getrf_f64_mgpu": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_int, c_i64, c_i64, c_p]),
    "rflu_getrf_f32_mgpu": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_int, c_i64, c_i64, c_p]),
    "rflu_mgpu_fill_uniform_f64": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_u64, c_dbl]),
    "rflu_mgpu_fill_uniform_f32": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_u64, c_dbl]),
    "rflu_profile_enable": (c_int, [c_p, c_int]),
    "rflu_profile_get": (c_int, [c_p, c_int, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64), ctypes.POINTER(c_dbl)]),
    "rflu_profile_get_bytes": (c_int, [c_p, c_int, ctypes.POINTER(c_dbl)]),
}

EXPORTS = dict(_PLAIN)
for _k, _v in _TYPED.items():
    for _s in ("f64", "f32"):
        EXPORTS[_k.format(s=_s)] = _v

K_GEMM, K_TRSM, K_LASWP, K_PANEL, K_TRANSPOSE, K_MISC = range(6)
KCLASS_NAMES = ["gemm", "trsm", "laswp", "panel", "transpose", "misc", "gemm_small", "laswp_wide"]
PATH_NONE, PATH_HIP_RECURSIVE, PATH_HIP_BLOCKED, PATH_HIP_LOOKAHEAD = 0, 1, 2, 3

_lib = None


def load() -> ctypes.CDLL:
    """dlopen librflu.so and declare every prototype.  Raises RfluError if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RfluError(
                f"{LIB_PATH} is missing: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback."
            )
        # One HIP/HSA runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64.  If librflu were
        # dlopen'ed first it would pull /opt/rocm's copies and the second HSA runtime loaded by torch finds no device.
        # Importing torch first lets librflu's NEEDED libamdhip64.so.7 resolve to the already-loaded one.
        try:
            import torch  # noqa: F401  (device-memory/stream plumbing only)
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            if not hasattr(lib, name) and os.environ.get("RFLU_LIB"):
                continue   # an older experiment build named by RFLU_LIB (scripts/r04_ab.sh): entry points added since are simply absent
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = load().rflu_last_error()
        raise RfluError(f"librflu status {status}: {msg.decode() if msg else ''}")


class Handle:
    """One device, one stream, reusable workspaces (the analogue of a LinearSolve cache)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.ptr = c_p()
        check(self.lib.rflu_create(ctypes.byref(self.ptr), int(device)))
        self.device = int(device)

    def close(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self.lib.rflu_destroy(self.ptr)
            self.ptr = c_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name: str, *args):
        check(getattr(self.lib, name)(self.ptr, *args))

    def last_path(self) -> int:
        return int(self.lib.rflu_last_path(self.ptr))

    def reload_tuning(self):
        """Read the RFLU_* tuning variables again (they are read once, when the handle is created)."""
        check(self.lib.rflu_reload_tuning(self.ptr))

    def set_stream(self, stream_ptr: int | None):
        check(self.lib.rflu_set_stream(self.ptr, c_p(stream_ptr or 0)))

    def synchronize(self):
        check(self.lib.rflu_synchronize(self.ptr))

    def update_stream(self) -> int:
        """hipStream_t (as int) of the CU-masked update stream."""
        out = c_p()
        check(self.lib.rflu_update_stream(self.ptr, ctypes.byref(out)))
        return int(out.value)

    def profile_enable(self, on):
        """False/0 = off, True/1 = synchronous per-launch timers (one-stream schedule), 2 = in-schedule event pairs."""
        check(self.lib.rflu_profile_enable(self.ptr, int(on)))

    def profile(self) -> dict:
        out = {}
        for k, name in enumerate(KCLASS_NAMES):
            ms, n, work = c_dbl(), c_i64(), c_dbl()
            check(self.lib.rflu_profile_get(self.ptr, k, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work)))
            by = c_dbl()
            check(self.lib.rflu_profile_get_bytes(self.ptr, k, ctypes.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "work": work.value, "bytes": by.value}
        return out


_handles: dict[int, Handle] = {}


def default_handle(device: int = 0) -> Handle:
    if device not in _handles:
        _handles[device] = Handle(device)
    return _handles[device]


_mgpu_objects: "weakref.WeakSet" = None  # live MultiGPU objects (multigpu.py registers itself)


def register_mgpu(obj) -> None:
    global _mgpu_objects
    import weakref

    if _mgpu_objects is None:
        _mgpu_objects = weakref.WeakSet()
    _mgpu_objects.add(obj)


def reload_tuning() -> None:
    """The RFLU_* variables are read ONCE per handle, at creation; changing os.environ later is ignored until this is called.
    Every default handle and every live MultiGPU object's per-device handles read the environment again.  (A Handle built
    by the caller has its own ``reload_tuning()``.)"""
    for h in _handles.values():
        h.reload_tuning()
    for g in list(_mgpu_objects or ()):
        if getattr(g, "ptr", None) is not None and g.ptr.value:
            check(g.lib.rflu_mgpu_reload_tuning(g.ptr))
