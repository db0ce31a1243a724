"""1-D block-column LU over the GPUs of one node: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

The reference has no distributed path (zero collective call sites); this is the partition the
north star asks for.  Structure:
  * the n x n matrix is cut into block columns of ``block`` (a multiple of 64) columns; block b lives on rank
    ``b % world`` as part of that rank's row-major slab (all n rows x its local columns) -- cyclic, so the shrinking
    trailing matrix stays balanced;
  * per block column: the owner factors the tall panel with the single-GPU recursive path (``rflu_panel_rm_*``:
    cooperative leaf panels + TRSM/GEMM inside the panel), packs {L\\U panel, ipiv segment, info} and broadcasts it --
    the ONLY collective of the path, one message per block column;
  * every rank then applies the interchanges to its local columns (``rflu_laswp_rm_*``), solves the block row
    (``rflu_trsm_rm_*``) and updates its trailing columns (``rflu_gemm_rm_*``, MFMA).  No reductions are needed.
The kernels are reached through an ``ops`` object so that the orchestration (ownership, message contents, update order)
is exercised on CPU under gloo in tests/test_distributed.py with a NumPy stand-in for ``ops``; the product path uses
``HipOps`` (C ABI of librflu.so) only.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist

NB = 64


class HipOps:
    """The four kernels of the path on row-major device slabs, through the C ABI (include/rflu.h)."""

    def __init__(self, handle, sfx: str):
        self.h = handle
        self.sfx = sfx
        self.es = 8 if sfx == "f64" else 4

    def _p(self, t, off=0):
        return ctypes.c_void_p(t.data_ptr() + off * self.es)

    # ---- streams of the lookahead schedule (None on the CPU stand-in) ----
    def make_streams(self, device):
        """(update stream U: CU-masked, from librflu;  panel stream P: an ordinary side stream)."""
        U = torch.cuda.ExternalStream(self.h.update_stream(), device=device)
        P = torch.cuda.Stream(device=device)
        return U, P

    def use(self, stream):
        """Route the following kernel launches of the library to ``stream``."""
        self.h.set_stream(stream.cuda_stream if stream is not None else None)

    def panel(self, R, ld, m, r0, c0, w, ipiv, pivot) -> int:
        info = ctypes.c_int64(0)
        self.h.call(f"rflu_panel_rm_{self.sfx}_dev", m, r0, c0, w, self._p(R), ld, ctypes.c_void_p(ipiv.data_ptr()),
                    int(pivot), ctypes.byref(info))
        return int(info.value)

    def laswp(self, R, ld, m, c0, ncols, ipiv, k0, k1):
        if ncols > 0 and k1 > k0:
            self.h.call(f"rflu_laswp_rm_{self.sfx}_dev", self._p(R), ld, m, c0, ncols, ctypes.c_void_p(ipiv.data_ptr()), k0, k1)

    def trsm(self, n, nrhs, L, l_off, ldl, B, b_off, ldb):
        if n > 0 and nrhs > 0:
            self.h.call(f"rflu_trsm_rm_{self.sfx}_dev", n, nrhs, self._p(L, l_off), ldl, self._p(B, b_off), ldb)

    def gemm(self, M, N, K, A, a_off, lda, B, b_off, ldb, C, c_off, ldc):
        if M > 0 and N > 0 and K > 0:
            self.h.call(f"rflu_gemm_rm_{self.sfx}_dev", M, N, K, self._p(A, a_off), lda, self._p(B, b_off), ldb,
                        self._p(C, c_off), ldc)

    def fill(self, R, ld, m, w, c0, seed, n_global, j0, diag_add=0.0):
        self.h.call(f"rflu_fill_uniform_{self.sfx}_dev", self._p(R, c0), m, w, ld, 1, seed, n_global, 0, j0, float(diag_add))


def block_layout(n: int, block: int, world: int, run: int = 1):
    """[(global col start, width, owner rank, local col offset on the owner)] for every block column.

    ``run`` consecutive block columns share an owner (owner = (b // run) % world): inside a run the owner goes from one
    panel to the next without waiting for any broadcast, so only every ``run``-th broadcast is on the critical chain."""
    out = []
    nblocks = (n + block - 1) // block
    local_off = [0] * world
    for b in range(nblocks):
        j0 = b * block
        w = min(block, n - j0)
        owner = (b // max(run, 1)) % world
        out.append((j0, w, owner, local_off[owner]))
        local_off[owner] += w
    return out, local_off


class BlockColumnLU:
    """Factor an n x n matrix distributed by block columns.  ``factor()`` is one step of bench.py's --gpus N path."""

    def __init__(self, ops, n, dtype, rank, world, device, *, block=512, pivot=True, seed=12, diag_add=0.0, group=None,
                 always_broadcast=False, run=None):
        if block % NB:
            raise ValueError("block must be a multiple of 64")
        self.ops, self.n, self.rank, self.world, self.device = ops, n, rank, world, device
        self.block, self.pivot, self.seed, self.diag_add, self.group = block, pivot, seed, diag_add, group
        self.dtype = dtype
        self.collective = world > 1 or always_broadcast  # always_broadcast: exercise the RCCL calls with one rank
        if run is None:
            run = int(os.environ.get("RFLU_DIST_RUN", "1"))
        self.run = max(1, run)
        self.layout, local_cols = block_layout(n, block, world, self.run)
        self.n_loc = local_cols[rank]
        self.ld = max(16, (self.n_loc + 15) // 16 * 16)
        self.R = torch.zeros((n, self.ld), dtype=dtype, device=device)       # this rank's slab, row-major
        self.ipiv = torch.zeros(n, dtype=torch.int64, device=device)          # full pivot vector on every rank
        wmax = min(block, n)
        # one message per block column: [panel rows j0..n) x w | ipiv segment | info], all carried as the matrix dtype's
        # bytes would lose int64 pivots for Float32, so pivots travel in a second int64 message appended to the first
        self.pbuf = [torch.zeros(n * wmax, dtype=dtype, device=device) for _ in range(2)]       # double-buffered by parity
        self.meta = [torch.zeros(wmax + 1, dtype=torch.int64, device=device) for _ in range(2)]  # ipiv segment + info
        self.info_dev = torch.zeros((), dtype=torch.int64, device=device)
        self.info = 0
        # panels taller than this many rows need more than the 32 CUs the update stream leaves free (one 512-row
        # workgroup per CU): their owner factors them BEFORE its own bulk update instead of next to it
        self.tall_rows = 32 * 512

    # ---- input -------------------------------------------------------------------------------------------------------
    def regenerate(self):
        for (j0, w, owner, lc) in self.layout:
            if owner == self.rank:
                self.ops.fill(self.R, self.ld, self.n, w, lc, self.seed, self.n, j0, self.diag_add)

    def load_global(self, A):
        """Scatter a full (n x n) host matrix into the slabs (tests)."""
        for (j0, w, owner, lc) in self.layout:
            if owner == self.rank:
                self.R[:, lc:lc + w] = torch.as_tensor(np.ascontiguousarray(A[:, j0:j0 + w]), dtype=self.dtype).to(self.device)

    # ---- the factorization ---------------------------------------------------------------------------------------------
    def _local_ranges(self, j0, w, owner, lc):
        """(end of the local columns left of block column (j0,w), start of the local columns right of it)."""
        if self.rank == owner:
            return lc, lc + w
        rs = sum(ww for (jj, ww, oo, _) in self.layout if oo == self.rank and jj < j0)
        return rs, rs

    def _update(self, j0, w, pbuf, c0, ncols):
        """Apply block column (j0,w), held packed in ``pbuf``, to the local columns [c0, c0+ncols): interchanges,
        block-row solve against L11, Schur update with L21."""
        if ncols <= 0:
            return
        n, ld, ops = self.n, self.ld, self.ops
        if self.pivot:
            ops.laswp(self.R, ld, n, c0, ncols, self.ipiv, j0, j0 + w)
        ops.trsm(w, ncols, pbuf, 0, w, self.R, j0 * ld + c0, ld)
        ops.gemm(n - j0 - w, ncols, w, pbuf, w * w, w, self.R, j0 * ld + c0, ld, self.R, (j0 + w) * ld + c0, ld)

    def factor(self):
        if os.environ.get("RFLU_DIST_SYNC") == "1" or len(self.layout) < 2:
            return self.factor_sync()
        return self.factor_lookahead()

    def factor_sync(self):
        """One block column at a time: panel -> broadcast -> update, everything on the current stream."""
        n, ld, ops = self.n, self.ld, self.ops
        self.info = 0
        self.info_dev.zero_()
        if not self.pivot:
            self.ipiv.copy_(torch.arange(1, n + 1, dtype=torch.int64, device=self.device))
        pbuf_all, meta = self.pbuf[0], self.meta[0]
        for (j0, w, owner, lc) in self.layout:
            rows = n - j0
            panel = pbuf_all[: rows * w].view(rows, w)
            if self.rank == owner:
                info = ops.panel(self.R, ld, n, j0, lc, w, self.ipiv, self.pivot)
                panel.copy_(self.R[j0:, lc:lc + w])
                meta[:w].copy_(self.ipiv[j0:j0 + w])
                meta[w] = info
            # ---- the one exchange step of the path: panel + pivots, owner -> everybody ----
            if self.collective:
                src = owner if self.group is None else dist.get_global_rank(self.group, owner)
                dist.broadcast(panel, src=src, group=self.group)
                dist.broadcast(meta[: w + 1], src=src, group=self.group)
            if self.rank != owner:
                self.ipiv[j0:j0 + w].copy_(meta[:w])
            # first non-zero info wins; kept on the device so that no block column forces a host synchronisation
            self.info_dev.copy_(torch.where(self.info_dev == 0, meta[w], self.info_dev))
            left_end, right_start = self._local_ranges(j0, w, owner, lc)
            if self.pivot:
                ops.laswp(self.R, ld, n, 0, left_end, self.ipiv, j0, j0 + w)
            self._update(j0, w, pbuf_all, right_start, self.n_loc - right_start)
        self.info = int(self.info_dev.item())
        return self.info

    def factor_lookahead(self):
        """Same operations, one block column of lookahead on two streams:
             U (CU-masked update stream): per block column b  [wait recv_b]  next-owner slice first, then the bulk update
             P (panel stream)          : owner of b+1 factors its block column as soon as its slice is updated, packs it,
                                         and EVERY rank issues the broadcast of b+1 on P -- while U still runs update b.
           Every rank issues the collectives in the same order (b = 0, 1, 2, ...); buffers are double-buffered by parity."""
        n, ld, ops = self.n, self.ld, self.ops
        gpu = self.device.type == "cuda"
        if gpu:
            U, P = ops.make_streams(self.device)
            cur = torch.cuda.current_stream(self.device)
            U.wait_stream(cur)
            P.wait_stream(cur)
        else:
            U = P = None

        class _On:  # `with _On(stream):` = torch stream context + library stream; a no-op on the CPU stand-in
            def __init__(s2, st): s2.st = st
            def __enter__(s2):
                if gpu:
                    s2.ctx = torch.cuda.stream(s2.st); s2.ctx.__enter__(); ops.use(s2.st)
            def __exit__(s2, *a):
                if gpu:
                    s2.ctx.__exit__(*a)

        def record(st):
            return st.record_event() if gpu else None

        def wait(st, ev):
            if gpu and ev is not None:
                st.wait_event(ev)

        self.info = 0
        with _On(U):
            self.info_dev.zero_()
            if not self.pivot:
                self.ipiv.copy_(torch.arange(1, n + 1, dtype=torch.int64, device=self.device))
        ev0 = record(U)
        wait(P, ev0)
        nb = len(self.layout)
        packed = [None] * nb   # event on P: block column b sits packed in its buffer on its owner
        works = [None] * nb    # the asynchronous broadcasts of block column b (panel, pivots)
        done = [None] * nb

        def await_works(b):
            """Make the CURRENT stream (the host, on the CPU stand-in) wait for block column b's broadcasts."""
            if works[b] is not None:
                for wk in works[b]:
                    wk.wait()

        def produce(b, ev_ready):
            """On P: factor + pack (owner), then EVERY rank issues the broadcast of block column b -- asynchronously: P
            goes on to its next panel while the transport runs; consumers wait for it where they need the data."""
            j0, w, owner, lc = self.layout[b]
            rows = n - j0
            pb, mt = self.pbuf[b % 2], self.meta[b % 2]
            panel = pb[: rows * w].view(rows, w)
            with _On(P):
                if b >= 2:
                    wait(P, done[b - 2])          # the buffer was last read by update b-2 ...
                    await_works(b - 2)            # ... and (on its sender) by broadcast b-2
                if self.rank == owner:
                    wait(P, ev_ready)             # this rank's slice has received update b-1
                    info = ops.panel(self.R, ld, n, j0, lc, w, self.ipiv, self.pivot)
                    panel.copy_(self.R[j0:, lc:lc + w])
                    mt[:w].copy_(self.ipiv[j0:j0 + w])
                    mt[w] = info
                    packed[b] = record(P)
                if self.collective:
                    src = owner if self.group is None else dist.get_global_rank(self.group, owner)
                    works[b] = [dist.broadcast(panel, src=src, group=self.group, async_op=True),
                                dist.broadcast(mt[: w + 1], src=src, group=self.group, async_op=True)]

        produce(0, ev0)
        for b in range(nb):
            j0, w, owner, lc = self.layout[b]
            pb, mt = self.pbuf[b % 2], self.meta[b % 2]
            left_end, right_start = self._local_ranges(j0, w, owner, lc)
            nxt_slice = 0
            ev_ready = None
            with _On(U):
                if self.rank == owner:
                    wait(U, packed[b])            # the owner reads its own packed copy: no need to wait for the transport
                else:
                    await_works(b)
                    self.ipiv[j0:j0 + w].copy_(mt[:w])
                self.info_dev.copy_(torch.where(self.info_dev == 0, mt[w], self.info_dev))
                if b + 1 < nb and self.rank == self.layout[b + 1][2]:
                    # this rank owns block column b+1: bring exactly those columns up to date first
                    nxt_slice = self.layout[b + 1][1]
                    assert self.layout[b + 1][3] == right_start
                    self._update(j0, w, pb, right_start, nxt_slice)
                    ev_ready = record(U)
                tall_next = (b + 1 < nb and nxt_slice > 0 and n - self.layout[b + 1][0] > self.tall_rows)
            if tall_next:
                # this rank owns a TALL block column b+1: it cannot run next to the update (not enough free CUs), and it is
                # what every other rank will wait for -- factor it first, whole GPU, then catch up on the bulk of update b
                # while the others are already applying b+1
                produce(b + 1, ev_ready)
            with _On(U):
                if tall_next:
                    wait(U, packed[b + 1])
                # (otherwise) the bulk of update b is queued BEFORE the host-blocking panel call, so the two overlap
                if self.pivot:
                    ops.laswp(self.R, ld, n, 0, left_end, self.ipiv, j0, j0 + w)
                self._update(j0, w, pb, right_start + nxt_slice, self.n_loc - right_start - nxt_slice)
                done[b] = record(U)
            if b + 1 < nb and not tall_next:
                produce(b + 1, ev_ready)
        for b in range(max(0, nb - 2), nb):   # nobody waited for the last sends on their sender
            with _On(P):
                await_works(b)
        if gpu:
            cur.wait_stream(U)
            cur.wait_stream(P)
            ops.use(None)
        self.info = int(self.info_dev.item())
        return self.info

    # ---- results -------------------------------------------------------------------------------------------------------
    def gather_factors(self):
        """Full packed L\\U on every rank as a host array (tests / small sizes only)."""
        full = torch.zeros((self.n, self.n), dtype=self.dtype, device=self.device)
        for (j0, w, owner, lc) in self.layout:
            blk = torch.zeros((self.n, w), dtype=self.dtype, device=self.device)
            if owner == self.rank:
                blk.copy_(self.R[:, lc:lc + w])
            if self.collective:
                src = owner if self.group is None else dist.get_global_rank(self.group, owner)
                dist.broadcast(blk, src=src, group=self.group)
            full[:, j0:j0 + w] = blk
        return full.cpu().numpy()

    def matvec_residual(self, trials: int = 2) -> float:
        """max over random x of ||P*A*x - L*(U*x)|| / ||A*x||, A regenerated from the seed; O(n^2) per rank.
        Checker for sizes where the n^3 residual is not affordable (uses torch ops, not the product's kernels)."""
        n, dev = self.n, self.device
        LU = self.R[:, : self.n_loc].clone()
        self.regenerate()
        A = self.R[:, : self.n_loc].clone()
        self.R[:, : self.n_loc] = LU
        cols = torch.cat([torch.arange(j0, j0 + w, device=dev) for (j0, w, o, _) in self.layout if o == self.rank]) \
            if self.n_loc else torch.zeros(0, dtype=torch.int64, device=dev)
        rows = torch.arange(n, device=dev)[:, None]
        Uloc = torch.where(rows <= cols[None, :], LU, torch.zeros_like(LU))
        Lloc = torch.where(rows > cols[None, :], LU, torch.zeros_like(LU))
        perm = np.arange(n)
        for i, t in enumerate(self.ipiv.cpu().numpy()):
            j = int(t) - 1
            if j != i:
                perm[i], perm[j] = perm[j], perm[i]
        perm = torch.from_numpy(perm).to(dev)
        worst = 0.0
        gen = torch.Generator(device="cpu").manual_seed(1234)
        for _ in range(trials):
            x = torch.rand(n, dtype=torch.float64, generator=gen).to(dev).to(self.dtype)
            xl = x[cols]
            ax = A.to(torch.float64) @ xl.to(torch.float64)
            ux = Uloc.to(torch.float64) @ xl.to(torch.float64)
            if self.collective:
                dist.all_reduce(ax, group=self.group)
                dist.all_reduce(ux, group=self.group)
            lz = Lloc.to(torch.float64) @ ux[cols] + torch.zeros(n, dtype=torch.float64, device=dev)
            if self.collective:
                dist.all_reduce(lz, group=self.group)
            lz = lz + ux  # unit diagonal of L
            r = torch.linalg.norm(ax[perm] - lz) / torch.linalg.norm(ax)
            worst = max(worst, float(r.item()))
        return worst
