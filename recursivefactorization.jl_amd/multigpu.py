"""Host-side wrapper of the multi-GPU C entry (``rflu_mgpu_*`` / ``rflu_getrf_*_mgpu``, include/rflu.h): ONE process drives
the GPUs of a node through librflu.so -- 1-D block-column slabs, one ``ncclBroadcast`` of {panel, pivots} per block column on
the library's own streams (RCCL over xGMI).  Naming one physical device several times ("fake multi-GPU") runs the same
partition logic with device-to-device copies instead of the broadcast: that is what a single-GPU box can test.

torch only allocates the slabs and moves test data; every compute call goes through the C ABI."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _ffi
from .distributed import block_layout


class MultiGPU:
    def __init__(self, devs):
        self.lib = _ffi.load()
        self.devs = [int(d) for d in devs]
        self.ndev = len(self.devs)
        self.ptr = ctypes.c_void_p()
        arr = (ctypes.c_int * self.ndev)(*self.devs)
        _ffi.check(self.lib.rflu_mgpu_create(ctypes.byref(self.ptr), self.ndev, arr))
        self.fake = bool(self.lib.rflu_mgpu_is_fake(self.ptr))
        _ffi.register_mgpu(self)   # _ffi.reload_tuning() reaches the per-device handles too

    @property
    def collectives(self) -> int:
        """ncclBroadcast calls enqueued so far (0 in fake mode; see RFLU_MGPU_FORCE_RCCL in include/rflu.h)."""
        return int(self.lib.rflu_mgpu_collectives(self.ptr))

    def close(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self.lib.rflu_mgpu_destroy(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- slabs ---------------------------------------------------------------------------------------------------------
    def alloc(self, n, dtype=torch.float64, block=512, run=1):
        """Row-major slabs (n x local columns, ld rounded up to 16 elements), one per logical device."""
        layout, local_cols = block_layout(n, block, self.ndev, run)
        slabs, lds = [], []
        for d in range(self.ndev):
            nl = int(self.lib.rflu_mgpu_local_cols(n, block, self.ndev, run, d))
            assert nl == local_cols[d], (nl, local_cols[d])
            ld = max(16, (nl + 15) // 16 * 16)
            slabs.append(torch.zeros((n, ld), dtype=dtype, device=f"cuda:{self.devs[d]}"))
            lds.append(ld)
        return slabs, lds, layout

    def _args(self, slabs, lds):
        P = (ctypes.c_void_p * self.ndev)(*[s.data_ptr() for s in slabs])
        L = (ctypes.c_int64 * self.ndev)(*lds)
        return P, L

    @staticmethod
    def _sfx(dtype):
        return "f64" if dtype == torch.float64 else "f32"

    def fill_uniform(self, n, slabs, lds, block, run, seed=12, diag_add=0.0):
        P, L = self._args(slabs, lds)
        fn = getattr(self.lib, f"rflu_mgpu_fill_uniform_{self._sfx(slabs[0].dtype)}")
        _ffi.check(fn(self.ptr, n, P, L, block, run, ctypes.c_uint64(seed), ctypes.c_double(diag_add)))

    def scatter(self, A, slabs, layout):
        """Full host matrix -> slabs (tests)."""
        for (j0, w, owner, lc) in layout:
            slabs[owner][:, lc:lc + w] = torch.as_tensor(np.ascontiguousarray(A[:, j0:j0 + w]), dtype=slabs[owner].dtype).to(slabs[owner].device)

    def gather(self, slabs, layout, n):
        """Slabs -> full packed L\\U on the host (tests / small sizes only)."""
        out = np.zeros((n, n), dtype=np.float64 if slabs[0].dtype == torch.float64 else np.float32)
        for (j0, w, owner, lc) in layout:
            out[:, j0:j0 + w] = slabs[owner][:, lc:lc + w].cpu().numpy()
        return out

    def getrf(self, n, slabs, lds, block, run=1, pivot=True):
        """Factor in place; returns (ipiv [host int64, 1-based, global rows] or None for NoPivot, info)."""
        P, L = self._args(slabs, lds)
        ipiv = np.zeros(n, dtype=np.int64)
        info = ctypes.c_int64(0)
        fn = getattr(self.lib, f"rflu_getrf_{self._sfx(slabs[0].dtype)}_mgpu")
        _ffi.check(fn(self.ptr, n, P, L, ctypes.c_void_p(ipiv.ctypes.data), int(bool(pivot)), block, run, ctypes.byref(info)))
        self.last_ipiv = ipiv
        return ipiv, int(info.value)
