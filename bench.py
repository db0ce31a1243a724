#!/usr/bin/env python
"""bench.py -- LU GFLOP/s (2n^3/3) of the MI355X-native recursive LU, the metric of BASELINE.json.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size SIZE] [--dtype f64|f32] [--nopivot] [--blocksize B]

A "step" is one factorization  lu!(A, ipiv)  of a dense uniform [0,1) n x n matrix that is already resident in HBM in
the reference's column-major layout; the result (packed L\\U, ipiv) is left in HBM.  lu! works in place, so the input is
regenerated on the device before every step -- INSIDE the timed region (0.34 ms at n = 16384); the K steps run in one
bracket (barrier + device synchronisation on both sides, max over ranks), so `value` = K * (2n^3/3) / t_bracket.

Workloads (BASELINE.json configs): 1 GPU -> n = 16384 (config 2, the one the 70 %-of-peak target is quoted on);
2 and 4 GPUs -> n = 32768 (config 3); 8 GPUs -> n = 65536 (config 4); --size overrides.  With N > 1 rank 0 also factors
the same n on ONE GPU in the same run (`one_gpu_same_n`, `speedup_vs_one_gpu`): the strong-scaling ratio the north star's
">= 6x at 8 GPUs over 1 GPU on N=65536" is defined on.

Extra objects on the JSON line:
  roofline     dominant kernel = the MFMA GEMM update (schur_complement!): algorithmic 2*M*N*K flops of every launch of
               one profiled factorization / their summed HIP-event durations, against the fp64 MFMA peak
  cpu_baseline the CPU restatement of the reference (oracle/, "port") built -march=native and timed on this host: threaded
               (thread = Val(true)) headline + serial figures at n = 512 (config 0) and n = 4096, on a bounded sample
  laswp        HBM rate of the row interchanges in the shipped schedule; sweep: block sizes 64/128/256 (config 2)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # dense MFMA peaks of MI355X (fp64: 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
DEFAULT_N = {1: 16384, 2: 32768, 4: 32768, 8: 65536}
SEED = 12  # Random.seed!(12), test/runtests.jl:9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", "--n", dest="n", type=int, default=0,
                    help="matrix size (use --size under torchrun: its own parser treats --n as ambiguous)")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--nopivot", action="store_true")
    ap.add_argument("--blocksize", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-one-gpu", action="store_true", help="N > 1: skip the same-n factorization on one GPU (speedup_vs_one_gpu)")
    ap.add_argument("--no-extras", action="store_true", help="skip the in-schedule timer pass and the block-size sweep (profiling runs)")
    ap.add_argument("--cpu-n", type=int, default=6144, help="size of the bounded CPU-baseline sample (~10-30 s of CPU work)")
    ap.add_argument("--block", type=int, default=512, help="block-column width of the multi-GPU layout")
    return ap.parse_args()


def cpu_baseline(cpu_n: int):
    """Time the oracle (CPU restatement of the reference's lu!, same nsplit/blocksize/threshold) on this host: built with
    -march=native on this machine, serial (thread = Val(false)) and threaded (thread = Val(true): the reference threads
    apply_permutation!, the TRSM and schur_complement!; the panel stays serial).  Bounded to ~10-30 s of CPU work."""
    import numpy as np

    import oracle as O

    native = O.use_native()   # oracle/_native/: compiled here for this host's cores (falls back to the shipped x86-64-v3 build)
    cores = min(64, os.cpu_count() or 1)

    def timed(n, threads, reps=1):
        O.set_threads(threads)
        A = O.np_uniform(n, n, SEED)
        ts, ip = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            _, ip, _ = O.lu(A)
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        return round(2 * n**3 / 3 / dt / 1e9, 3), dt, ip

    O.set_threads(1)
    O.lu(O.np_uniform(512, 512, SEED))                      # warm-up
    n512, _, _ = timed(512, 1, reps=5)                      # BASELINE config 0 verbatim: lu!(rand(512,512)), Float64, pivoted, serial
    s4096, _, _ = timed(4096, 1)                            # the size that overlaps the GPU runs (config 1), serial
    t4096, _, _ = timed(4096, cores)
    gf, dt, ipiv = timed(cpu_n, cores)                      # headline sample (also provides the pivots for the parity check)
    O.set_threads(1)
    res = {"value": gf, "unit": "GFLOP/s", "cores": cores, "kind": "port",
           "sample": f"one lu!(A, thread = Val(true)) of the n={cpu_n} Float64 uniform matrix (seed {SEED}) by oracle/rflu_oracle.c "
                     f"(OpenMP over the reference's threaded loops, {cores} threads) in {dt:.2f} s",
           "serial_n512_gflops": n512, "serial_n4096_gflops": s4096, "threaded_n4096_gflops": t4096,
           "build": "-O3 -march=native on this host" if native else "-O3 -march=x86-64-v3 (shipped build)",
           "host_cpus": os.cpu_count()}
    try:  # context only: LAPACK getrf on all host cores
        import scipy.linalg as sla

        A = O.np_uniform(cpu_n, cpu_n, SEED)
        t0 = time.perf_counter()
        sla.lapack.dgetrf(A)
        res["lapack_getrf_allcores_gflops"] = round(2 * cpu_n**3 / 3 / (time.perf_counter() - t0) / 1e9, 1)
    except Exception:
        pass
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    res["host_cpu"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return res, ipiv


def mgpu_matvec_residual(mg, n, slabs, lds, layout, block, run, pivot, trials=2):
    """max over random x of ||P*A*x - L*(U*x)|| / ||A*x|| for factors held in block-column slabs (one per device), A
    regenerated from the seed into scratch slabs; O(n^2) per device, plain torch ops as an independent checker."""
    import numpy as np
    import torch

    ipiv, _ = None, None
    dev0 = slabs[0].device
    As, _, _ = mg.alloc(n, slabs[0].dtype, block, run)
    mg.fill_uniform(n, As, lds, block, run, seed=SEED)
    # the pivots live on the host after getrf: factor() returned them; recompute the permutation from the last call
    ipiv = mg.last_ipiv
    perm = np.arange(n)
    if pivot:
        for i, t in enumerate(ipiv):
            j = int(t) - 1
            if j != i:
                perm[i], perm[j] = perm[j], perm[i]
    perm = torch.from_numpy(perm).to(dev0)
    cols = []
    for d in range(mg.ndev):
        cs = [torch.arange(j0, j0 + w) for (j0, w, o, lc) in layout if o == d]
        cols.append(torch.cat(cs).to(slabs[d].device) if cs else torch.zeros(0, dtype=torch.int64, device=slabs[d].device))
    gen = torch.Generator(device="cpu").manual_seed(1234)
    worst = 0.0
    for _ in range(trials):
        x = torch.rand(n, dtype=torch.float64, generator=gen)
        ax = torch.zeros(n, dtype=torch.float64, device=dev0)
        ux = torch.zeros(n, dtype=torch.float64, device=dev0)
        for d in range(mg.ndev):
            nl = cols[d].numel()
            if nl == 0:
                continue
            dd = slabs[d].device
            xl = x.to(dd)[cols[d]]
            rows = torch.arange(n, device=dd)[:, None]
            ax += (As[d][:, :nl].to(torch.float64) @ xl).to(dev0)
            LUd = slabs[d][:, :nl].to(torch.float64)
            ux += (torch.where(rows <= cols[d][None, :], LUd, torch.zeros((), dtype=torch.float64, device=dd)) @ xl).to(dev0)
            del LUd
        lz = ux.clone()
        for d in range(mg.ndev):
            nl = cols[d].numel()
            if nl == 0:
                continue
            dd = slabs[d].device
            rows = torch.arange(n, device=dd)[:, None]
            LUd = slabs[d][:, :nl].to(torch.float64)
            lz += (torch.where(rows > cols[d][None, :], LUd, torch.zeros((), dtype=torch.float64, device=dd)) @ ux.to(dd)[cols[d]]).to(dev0)
            del LUd
        r = torch.linalg.norm(ax[perm] - lz) / torch.linalg.norm(ax)
        worst = max(worst, float(r.item()))
    return worst


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    from recursivefactorization.jl_amd import _ffi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # RFLU_BENCH_ONE_GPU=1 (+ RFLU_BENCH_BACKEND=gloo): every rank on device 0 -- a functional rehearsal of the multi-rank
    # path on a single-GPU box (RCCL refuses two ranks on one device); never a performance number
    one_gpu = os.environ.get("RFLU_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("RFLU_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with a single rank (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("RFLU_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = args.n or DEFAULT_N.get(args.gpus, 16384 * max(1, args.gpus // 2))
    sfx = args.dtype
    tdt = torch.float64 if sfx == "f64" else torch.float32
    pivot = 0 if args.nopivot else 1
    h = _ffi.Handle(local_rank)
    h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    flops = 2.0 * n**3 / 3.0

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1 or force_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    single = world == 1 and not force_dist
    if single:
        A = torch.empty((n, n), dtype=tdt, device=dev)  # memory = column-major n x n (lda = n)
        ipiv = torch.empty(n, dtype=torch.int64, device=dev)
        info = ctypes.c_int64(0)

        def regenerate():
            h.call(f"rflu_fill_uniform_{sfx}_dev", ctypes.c_void_p(A.data_ptr()), n, n, n, 0, SEED, n, 0, 0, 0.0)

        def step():
            h.call(f"rflu_getrf_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ipiv.data_ptr()),
                   pivot, args.blocksize, ctypes.byref(info))
    else:
        from recursivefactorization.jl_amd import distributed as D

        # runs of 4 consecutive block columns per owner: 3 of 4 panel broadcasts overlap with the owner's next panel instead
        # of sitting on the critical chain; RFLU_DIST_RUN overrides
        run = int(os.environ.get("RFLU_DIST_RUN", "4" if world > 1 else "1"))
        # Default at N > 1: the multi-GPU C entry of librflu.so (rflu_getrf_*_mgpu): ONE process -- rank 0 -- drives all N
        # GPUs, ncclBroadcast of {panel, pivots} on the library's own streams, no host synchronisation per block column;
        # the other ranks only take part in the barriers.  RFLU_BENCH_MGPU=python (or a failure to set the C path up)
        # selects the one-process-per-GPU torch.distributed driver (recursivefactorization.jl_amd/distributed.py).
        mgpu_mode = os.environ.get("RFLU_BENCH_MGPU", "c" if (world > 1 and not one_gpu) else "python")
        mg = None
        if mgpu_mode == "c":
            ok = torch.zeros(1, dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
            if rank == 0:
                try:
                    from recursivefactorization.jl_amd.multigpu import MultiGPU

                    # RFLU_BENCH_ONE_GPU rehearsal: the N logical devices all name GPU 0 ("fake multi-GPU": copies instead of
                    # the RCCL broadcast) -- functional only, never a performance number
                    mg = MultiGPU([0] * world if one_gpu else list(range(world)))
                    slabs, lds, mlayout = mg.alloc(n, tdt, args.block, run)
                    ok[0] = 1
                except Exception as exc:  # noqa: BLE001 -- reported below, on every rank
                    print(f"bench: multi-GPU C entry unavailable on rank 0: {exc}", file=sys.stderr)
                    mg = None
            dist.broadcast(ok, src=0)
            if int(ok.item()) != 1:
                # No silent change of what is measured: the C entry needs rank 0 to see all N devices (and librccl.so).  The
                # one-process-per-GPU torch.distributed driver is a different schedule; it runs only when asked for by name.
                raise SystemExit(f"bench.py --gpus {world}: rank 0 could not set up the multi-GPU C entry "
                                 f"(visible devices: {torch.cuda.device_count()}); set RFLU_BENCH_MGPU=python to time the "
                                 "torch.distributed driver instead")
        if mgpu_mode == "c":
            mg_info = [0]

            def regenerate():
                if rank == 0:
                    mg.fill_uniform(n, slabs, lds, args.block, run, seed=SEED)

            def step():
                if rank == 0:
                    _, mg_info[0] = mg.getrf(n, slabs, lds, args.block, run, pivot=bool(pivot))
            job = None
        else:
            job = D.BlockColumnLU(D.HipOps(h, sfx), n, tdt, rank, world, dev, block=args.block, pivot=bool(pivot), seed=SEED,
                                  always_broadcast=force_dist, run=run)
            regenerate = job.regenerate
            step = job.factor

    for _ in range(args.warmup):
        regenerate()
        barrier()
        step()
    # exactly K steps inside ONE bracket (barrier + synchronize on both sides).  lu! works in place, so every step first
    # refills its input on the device (a 0.34 ms kernel at n = 16384, < 0.4 % of a step) -- inside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        regenerate()
        step()
    barrier()
    total = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(total, op=dist.ReduceOp.MAX)
    total_s = float(total.item())
    ms_per_step = 1e3 * total_s / max(args.steps, 1)
    gflops = flops * args.steps / total_s / 1e9

    # ---- N > 1: the SAME n on ONE GPU in the same run (rank 0, device 0), so that the line carries the ratio the north star's
    # scaling target is defined on (">= 6x at 8 GPUs over 1 GPU on N=65536"): strong scaling, same matrix, same seed ----
    one_gpu_same_n = None
    if not single and world > 1:
        if rank == 0 and not args.no_one_gpu:
            A1 = torch.empty((n, n), dtype=tdt, device=dev)
            ip1 = torch.empty(n, dtype=torch.int64, device=dev)
            inf1 = ctypes.c_int64(0)
            ts = []
            for it in range(2):   # one warm-up (workspaces, stream placement), one timed
                h.call(f"rflu_fill_uniform_{sfx}_dev", ctypes.c_void_p(A1.data_ptr()), n, n, n, 0, SEED, n, 0, 0, 0.0)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                h.call(f"rflu_getrf_{sfx}_dev", n, n, ctypes.c_void_p(A1.data_ptr()), n, ctypes.c_void_p(ip1.data_ptr()),
                       pivot, args.blocksize, ctypes.byref(inf1))
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t1)
            one_ms = 1e3 * ts[-1]
            one_gpu_same_n = {"n": n, "ms": round(one_ms, 3), "gflops": round(flops / ts[-1] / 1e9, 2),
                              "frac_of_mfma_peak": round(flops / ts[-1] / 1e12 / PEAK_TFLOPS[sfx], 4), "info": int(inf1.value),
                              "note": "rank 0 factors the same n x n matrix (same generator and seed) on one GPU through "
                                      "rflu_getrf_*_dev, one warm-up + one timed factorization, outside the timed multi-GPU region"}
            if pivot and mgpu_mode == "c" and mg is not None and getattr(mg, "last_ipiv", None) is not None:
                one_gpu_same_n["ipiv_equal_to_multi_gpu"] = bool(np.array_equal(ip1.cpu().numpy(), mg.last_ipiv))
            del A1, ip1
            torch.cuda.empty_cache()
        barrier()

    # ---- roofline of the dominant kernel: one extra profiled factorization (HIP events on the launch stream) ----
    roof = None
    kern = {}
    if True:
        regenerate()
        barrier()
        # mode 3: the single-stream blocked schedule with an event pair around every launch and no host wait in between (mode 1 waits
        # for every launch: the GPU idles ~20 us behind each kernel and its power management answers with a lower clock)
        h.profile_enable(3 if single else True)
        if single:
            step()
        elif job is not None:
            job.factor_sync()   # one block column at a time on one stream: every launch bracketed by HIP events
        elif rank == 0:
            # C entry: the per-GPU kernels are those of the single-GPU path; profile one single-GPU factorization of the
            # 1-GPU workload size on device 0 (the multi-GPU handles keep no per-class timers)
            np1 = min(n, DEFAULT_N[1])
            Ap = torch.empty((np1, np1), dtype=tdt, device=dev)
            ipp = torch.empty(np1, dtype=torch.int64, device=dev)
            infp = ctypes.c_int64(0)
            h.call(f"rflu_fill_uniform_{sfx}_dev", ctypes.c_void_p(Ap.data_ptr()), np1, np1, np1, 0, SEED, np1, 0, 0, 0.0)
            h.call(f"rflu_getrf_{sfx}_dev", np1, np1, ctypes.c_void_p(Ap.data_ptr()), np1, ctypes.c_void_p(ipp.data_ptr()),
                   pivot, 0, ctypes.byref(infp))
            del Ap, ipp
        barrier()
        kern = h.profile()
        h.profile_enable(False)
        g = kern["gemm"]
        if g["launches"] > 0 and g["ms"] > 0:
            ach = g["work"] / (g["ms"] * 1e-3) / 1e12
            traffic = None
            traffic_note = "traffic: null -- no PMC pass of THIS build and workload under profiles/ (scripts/collect_profiles.sh)"
            try:  # HBM bytes per launch from the committed PMC passes of this exact workload AND build (profiles/), if present
                import importlib.util

                with open(os.path.join(ROOT, "profiles", "gemm_traffic.json")) as f:
                    tj = json.load(f)
                bspec = importlib.util.spec_from_file_location("_rflu_build", os.path.join(ROOT, "recursivefactorization.jl_amd", "build.py"))
                bmod = importlib.util.module_from_spec(bspec)
                bspec.loader.exec_module(bmod)
                if tj.get("n") == n and tj.get("dtype") == sfx and tj.get("sources_sha1") == bmod.sources_digest():
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_note = ("traffic = (2*FETCH_SIZE+WRITE_SIZE)*1024 per launch, separate rocprofv3 --pmc passes of this "
                                    "build: " + tj.get("source", "profiles/"))
            except (OSError, ValueError, KeyError):
                pass
            roof = {"bound": "mfma", "kernel": "gemm_sub_kernel (schur_complement!, C -= A*B)", "achieved": round(ach, 3),
                    "peak": PEAK_TFLOPS[sfx], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[sfx], 4), "traffic": traffic,
                    "launches": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                    "flops_per_launch": g["work"] / g["launches"],
                    "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                    "note": ("all gemm_sub_kernel launches of one profiled factorization (single-stream blocked schedule, a HIP "
                             "event pair around every launch on the launch stream, read after the factorization); " + traffic_note) if single else
                            ("rank 0's gemm_sub_kernel launches of one profiled single-stream factorization (per-GPU figure)"
                             if job is not None else
                             "per-GPU kernel: gemm_sub_kernel launches of one profiled single-GPU factorization of the 1-GPU "
                             "workload on device 0 (the multi-GPU run uses the same kernel on every device)")}

    # ---- the same kernels inside the shipped two-stream schedule (event pairs on the launch streams, nothing waited for):
    # what the GEMM achieves next to the critical-path stream, and the HBM rate of the row interchanges (laswp)
    laswp = None
    sweep = None
    if single and not args.no_extras:
        regenerate()
        barrier()
        h.profile_enable(2)
        step()
        barrier()
        ks = h.profile()
        h.profile_enable(False)
        through_engine = h.last_path() == 4   # RFLU_PATH_HIP_ENGINE: the trailing updates ran inside the resident engine_kernel
        g2, lw_small, lw_wide = ks["gemm"], ks["laswp"], ks["laswp_wide"]
        lw = {k: lw_small[k] + lw_wide[k] for k in ("ms", "launches", "work")}
        if roof is not None and g2["launches"] > 0 and g2["ms"] > 0:
            ach2 = g2["work"] / (g2["ms"] * 1e-3) / 1e12
            # the top-level achieved / frac are the SHIPPED schedule's figure; the single-stream profiled pass is kept beside it
            roof["achieved_profiled"] = roof["achieved"]
            roof["frac_profiled"] = roof["frac"]
            roof["avg_launch_ms_profiled"] = roof["avg_launch_ms"]
            roof["launches_profiled"] = roof["launches"]
            roof["achieved"] = round(ach2, 3)
            roof["frac"] = round(ach2 / PEAK_TFLOPS[sfx], 4)
            roof["avg_launch_ms"] = round(g2["ms"] / g2["launches"], 4)
            roof["launches"] = g2["launches"]
            roof["flops_per_launch"] = g2["work"] / g2["launches"]
            roof["algorithmic_bytes_per_launch"] = g2["bytes"] / g2["launches"]
            roof["achieved_in_schedule"] = round(ach2, 3)
            roof["frac_in_schedule"] = round(ach2 / PEAK_TFLOPS[sfx], 4)
            roof["launches_in_schedule"] = g2["launches"]
            roof["note_in_schedule"] = ("gemm_sub_kernel launches with K >= 256 of one factorization in the shipped schedule "
                                        "(block-column lookahead, then leaf-wise): the bulk updates on the CU-masked stream "
                                        "(224 of 256 CUs) next to the critical path, with the clock the power management grants "
                                        "after the light phases (DESIGN.md section 7); sum of flops / sum of launch durations")
            if through_engine:
                # the shipped schedule of this size runs its updates in ONE resident kernel (csrc/engine.hip: the persistent update engine,
                # default for pivoted matrices of more than 11264 columns): that launch is the dominant kernel
                roof["kernel"] = ("engine_kernel (persistent update engine: interchanges, block-row solves and every Schur tile "
                                  "C -= A*B of the factorization, pulled from per-column-block counters; the tile code is gemm_sub_kernel's)")
                roof["note_in_schedule"] = ("ONE launch: the engine is resident on 224 of 256 CUs for the whole factorization.  flops = the "
                                            "Schur updates it performs (sum of 2 M N K over its operations), duration = its residency (HIP "
                                            "event pair on its stream), i.e. waiting for the chain of leaves included -- RFLU_ENGINE_TRACE=1 "
                                            "splits the workgroups' time (profiles/: *_engine_workgroup_time.txt; DESIGN.md section 3: tiles 59 + 8 %, "
                                            "strips and solves 5 %, between units 28 %); the tile kernel by itself is frac_profiled")
                # HBM traffic of the resident kernel: rocprofv3 --pmc runs one kernel at a time, which a kernel that waits for the chain's
                # kernels cannot survive -- the counters are those of the engine REPLAYED ALONE on a factored image (RFLU_ENGINE_REPLAY=1:
                # the same operations on the same addresses with every leaf counted as done, scripts/pmc_engine.sh), committed under
                # profiles/ with the hash of the sources they were measured on
                roof["traffic_profiled"] = roof.get("traffic")
                roof["traffic"] = None
                etraffic_note = ("traffic: null -- no replayed-engine PMC pass of THIS build under profiles/ (scripts/pmc_engine.sh, "
                                 "scripts/make_engine_traffic.py)")
                try:
                    with open(os.path.join(ROOT, "profiles", "engine_traffic.json")) as f:
                        ej = json.load(f)
                    if ej.get("n") == n and ej.get("dtype") == sfx and ej.get("sources_sha1") == bmod.sources_digest():
                        roof["traffic"] = ej["hbm_bytes_per_launch"]
                        roof["traffic_ratio"] = round(ej["hbm_bytes_per_launch"] / roof["algorithmic_bytes_per_launch"], 3)
                        roof["mfma_busy_frac_replayed"] = ej.get("mfma_busy_frac")
                        roof["replayed_alone_tflops"] = ej.get("replay_alone_tflops")
                        etraffic_note = ("traffic = (2*FETCH_SIZE+WRITE_SIZE)*1024 of the ONE engine_kernel launch replayed alone (separate rocprofv3 "
                                         "--pmc passes of this build): " + ej.get("source", "profiles/"))
                except (OSError, ValueError, KeyError, NameError):
                    pass
                roof["note"] = (roof["note_in_schedule"] + "; " + etraffic_note + "; traffic_profiled = HBM bytes per launch of the tile "
                                "kernel in the profiled single-stream pass; " + roof["note"])
        if lw["launches"] > 0 and lw["ms"] > 0:
            tbs = lw["work"] / (lw["ms"] * 1e-3) / 1e12
            esz = 8 if sfx == "f64" else 4
            laswp = {"bound": "hbm", "kernel": "laswp_kernel (apply_permutation!, row interchanges on the row-major copy)",
                     "achieved": round(tbs * 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(tbs / 8.0, 4),
                     "launches": lw["launches"], "total_ms": round(lw["ms"], 3),
                     "algorithmic_bytes": lw["work"], "algorithmic_bytes_expected": 4.0 * esz * n * n,
                     "note": "all laswp launches of one factorization in the shipped schedule; algorithmic bytes = "
                             "4*sizeof(T) per pivot per column (two rows read + written), sum / sum of launch durations. "
                             "wide = the launches that move >= 32 MiB (trailing-update and finished-column interchanges): the "
                             "bandwidth-bound share; the rest are the per-leaf launches (a few MB each, latency-bound)"}
            if lw_wide["launches"] > 0 and lw_wide["ms"] > 0:
                tw = lw_wide["work"] / (lw_wide["ms"] * 1e-3) / 1e12
                laswp["wide"] = {"achieved": round(tw * 1e3, 1), "frac": round(tw / 8.0, 4), "launches": lw_wide["launches"],
                                 "total_ms": round(lw_wide["ms"], 3), "share_of_bytes": round(lw_wide["work"] / lw["work"], 3)}
        # the same kernel with the GPU to itself: one block column's 512 interchanges over all n columns (pivot rows uniform over
        # the rows below), ten launches, event pairs as above.  The in-schedule figure runs on the CU-masked stream next to the
        # chain's kernels and on column ranges that shrink with the trailing matrix; this one says what the kernel can do.
        if laswp is not None and pivot and n >= 1024:
            npv = min(512, n // 2)
            ip2 = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
            ip2[:npv] = torch.randint(npv, n, (npv,), device=dev, generator=torch.Generator(device=dev).manual_seed(7)) + 1
            call = lambda: h.call(f"rflu_laswp_rm_{sfx}_dev", ctypes.c_void_p(A.data_ptr()), n, n, 0, n,
                                  ctypes.c_void_p(ip2.data_ptr()), 0, npv)
            call(); barrier()
            h.profile_enable(2)
            for _ in range(10):
                call()
            barrier()
            ka = h.profile()
            h.profile_enable(False)
            la = {k: ka["laswp"][k] + ka["laswp_wide"][k] for k in ("ms", "launches", "work")}
            if la["launches"] > 0 and la["ms"] > 0:
                ta = la["work"] / (la["ms"] * 1e-3) / 1e12
                laswp["alone"] = {"achieved": round(ta * 1e3, 1), "frac": round(ta / 8.0, 4), "launches": la["launches"],
                                  "avg_launch_us": round(la["ms"] * 1e3 / la["launches"], 2),
                                  "workload": f"{npv} interchanges x {n} columns, nothing else on the GPU"}
        if laswp is not None and through_engine:
            # Inside the shipped schedule of this size the interchanges are not launches of their own: the engine applies them in its
            # strip units (in front of every block-row solve) and as deferred-interchange units on the finished columns; the launches
            # timed above are the chain's few-MB per-leaf ones only.  The in-schedule HBM rate of laswp_kernel therefore does not exist
            # here: the headline figure of this object is the kernel BY ITSELF (`alone`), next to the PMC traffic ratio of the stream
            # schedule's launches (profiles/: 8.39 GB moved for 8.59 GB algorithmic, no re-reads) and what the engine's own
            # deferred-interchange units do per workgroup.
            ins = {k: laswp.pop(k) for k in ("achieved", "frac", "launches", "total_ms", "algorithmic_bytes") if k in laswp}
            laswp.pop("wide", None)
            laswp["chain_launches_only"] = {"gb_per_s": ins.get("achieved"), "launches": ins.get("launches"), "total_ms": ins.get("total_ms"),
                                            "algorithmic_bytes": ins.get("algorithmic_bytes"),
                                            "note": "the per-leaf launches of the critical path (latency-bound, a few MB each): NOT the factorization's "
                                                    "interchange traffic, which the engine moves"}
            if "alone" in laswp:
                laswp["achieved"] = laswp["alone"]["achieved"]
                laswp["frac"] = laswp["alone"]["frac"]
            laswp["note"] = ("engine schedule: achieved / frac = laswp_kernel with the GPU to itself (`alone`); in the schedule the interchanges "
                             "run inside engine_kernel (strip + deferred-interchange units), see engine_deferred_units")
            try:   # one traced factorization: workgroup time of the engine's deferred-interchange units (pure row moves)
                os.environ["RFLU_ENGINE_TRACE"] = "1"
                h.reload_tuning()
                regenerate(); barrier(); step(); barrier()
                acct = (ctypes.c_longlong * 8)()
                h.call("rflu_debug_engine_acct", acct)
                esz = 8 if sfx == "f64" else 4
                W = 512
                nb = (n + W - 1) // W
                left_bytes = 4.0 * esz * sum(W * (b * W) for b in range(nb)) + 4.0 * esz * nb * sum((W // 64 - 1 - u) * 64 * 64 for u in range(W // 64))
                wg_s = acct[3] / 1e8
                if wg_s > 0:
                    laswp["engine_deferred_units"] = {"algorithmic_bytes": left_bytes, "workgroup_ms": round(wg_s * 1e3, 2),
                                                      "gb_per_s_per_workgroup": round(left_bytes / wg_s / 1e9, 2),
                                                      "share_of_all_interchange_bytes": round(left_bytes / (4.0 * esz * n * n), 3),
                                                      "note": "RFLU_ENGINE_TRACE=1: time the engine's workgroups spend in deferred-interchange "
                                                              "units (the interchanges of later block columns on the finished columns to the left: "
                                                              "half of all interchange bytes), summed over the 448 resident workgroups"}
            except Exception as e:   # measurement aid only
                laswp["engine_deferred_units"] = {"error": str(e)}
            finally:
                os.environ.pop("RFLU_ENGINE_TRACE", None)
                h.reload_tuning()
        # BASELINE config 2: the block-size sweep 64 / 128 / 256 at this size (two timed factorizations each)
        if pivot and args.blocksize == 0 and n >= 1024:
            sweep = []
            for bs in (64, 128, 256):
                def step_bs(bs=bs):
                    h.call(f"rflu_getrf_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ipiv.data_ptr()),
                           pivot, bs, ctypes.byref(info))
                regenerate(); barrier(); step_bs()
                barrier()
                ts0 = time.perf_counter()
                for _ in range(2):
                    regenerate()
                    step_bs()
                barrier()
                dt = (time.perf_counter() - ts0) / 2
                sweep.append({"blocksize": bs, "ms": round(1e3 * dt, 3), "gflops": round(flops / dt / 1e9, 1),
                              "frac_of_mfma_peak": round(flops / dt / 1e12 / PEAK_TFLOPS[sfx], 4)})

    # ---- variants of the same workload (extra keys, never `value`): NoPivot, and the reference's butterfly route
    # (src/butterflylu.jl:45-55): ONE pass A <- U'AV (HBM-bound), then lu!(A, Val(false)) -- no pivot search, hence no per-column
    # latency chain -- and x = V ((U'AV) \ (U'b)) with the outer products as O(n) butterflies
    variants = None
    if single and not args.no_extras and pivot and args.blocksize == 0 and n % 4 == 0 and n >= 1024:
        from recursivefactorization.jl_amd import butterfly as BF

        esz = 8 if sfx == "f64" else 4
        uv = torch.from_numpy(BF.generate_random(n, np.float64 if sfx == "f64" else np.float32, 888)).to(dev)
        Acm = A.T   # the logical n x n matrix as a column-major view (stride(0) == 1) of the same memory

        def step_np():
            h.call(f"rflu_getrf_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(0), 0, 0, ctypes.byref(info))

        def timed(fn, reps=3):
            regenerate(); fn(); barrier()
            tt0 = time.perf_counter()
            for _ in range(reps):
                regenerate()
                fn()
            barrier()
            return (time.perf_counter() - tt0) / reps

        def step_bf():
            BF.butterfly_mul_(Acm, uv, handle=h)
            step_np()

        t_np = timed(step_np)
        t_bf = timed(step_bf)
        # the butterfly pass alone: HIP events on the launch stream (h runs on torch's current stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        regenerate(); barrier()
        e0.record()
        for _ in range(5):
            BF.butterfly_mul_(Acm, uv, handle=h)
        e1.record()
        barrier()
        bf_ms = e0.elapsed_time(e1) / 5
        # backward error of the solve through the butterfly route (A x = b, b = A * ones)
        regenerate(); barrier()
        A0 = A.clone()
        bvec = A0.T @ torch.ones(n, dtype=tdt, device=dev)
        step_bf(); barrier()
        rhs = bvec.clone()
        h.call(f"rflu_butterfly_vec_{sfx}_dev", n, 1, ctypes.c_void_p(rhs.data_ptr()), n, ctypes.c_void_p(uv.data_ptr()), 1)   # U' b
        h.call(f"rflu_getrs_{sfx}_dev", n, 1, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(0), ctypes.c_void_p(rhs.data_ptr()), n)
        h.call(f"rflu_butterfly_vec_{sfx}_dev", n, 1, ctypes.c_void_p(rhs.data_ptr()), n, ctypes.c_void_p(uv.data_ptr()), 0)   # V y
        barrier()
        r = A0.T.to(torch.float64) @ rhs.to(torch.float64) - bvec.to(torch.float64)
        berr = float((torch.linalg.norm(r) / (torch.linalg.norm(A0.to(torch.float64)) * torch.linalg.norm(rhs.to(torch.float64)))).item())
        del A0
        # the solve step ldiv!(F, b) for ONE right-hand side on pivoted factors (what LinearSolve calls right after lu!): both
        # triangles + the interchanges of b, HIP events on the launch stream
        ipd = ipiv if pivot else None
        bsol = torch.rand(n, dtype=tdt, device=dev)
        bkeep = bsol.clone()
        def time_solve(entry):
            def solve_once():
                h.call(entry, n, 1, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ipd.data_ptr() if ipd is not None else 0),
                       ctypes.c_void_p(bsol.data_ptr()), n if "rm" not in entry else 16)
            solve_once(); barrier()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(5):
                bsol.copy_(bkeep)
                solve_once()
            s1.record()
            barrier()
            return s0.elapsed_time(s1) / 5
        regenerate(); step(); barrier()
        solve_cm_ms = time_solve(f"rflu_getrs_{sfx}_dev")          # column-major factors: + one transpose of F into the row-major workspace
        solve_ms = None
        solve64_ms = None
        if n % 16 == 0:
            regenerate(); barrier()
            h.call(f"rflu_getrf_rm_{sfx}_dev", n, n, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ipiv.data_ptr()), 1, 0, ctypes.byref(info))
            bsol16 = torch.zeros((n, 16), dtype=tdt, device=dev)   # row-major n x 1 right-hand side with leading dimension 16
            bsol16[:, 0] = bkeep
            bkeep, bsol = bsol16.clone(), bsol16
            solve_ms = time_solve(f"rflu_getrs_rm_{sfx}_dev")      # the factors as the library keeps them: the solve kernels alone
            # a BLOCK of 64 right-hand sides on the same factors (trsv.hip: trsm_chain_kernel, two chains of 32 columns on the MFMA units)
            b64 = torch.rand((n, 64), dtype=tdt, device=dev)
            b64k = b64.clone()
            def solve64_once():
                h.call(f"rflu_getrs_rm_{sfx}_dev", n, 64, ctypes.c_void_p(A.data_ptr()), n, ctypes.c_void_p(ipd.data_ptr() if ipd is not None else 0),
                       ctypes.c_void_p(b64.data_ptr()), 64)
            solve64_once(); barrier()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(5):
                b64.copy_(b64k)
                solve64_once()
            s1.record()
            barrier()
            solve64_ms = s0.elapsed_time(s1) / 5
            del b64, b64k
        variants = {
            "solve_ms": round(solve_ms, 3) if solve_ms is not None else None,
            "solve_cm_ms": round(solve_cm_ms, 3),
            "solve64_ms": round(solve64_ms, 3) if solve64_ms is not None else None,
            "solve_note": "ldiv!(F, b), one right-hand side, pivoted factors: interchanges of b + L and U solves (csrc/trsv.hip: one "
                          "cooperative launch per triangle).  solve_ms: rflu_getrs_rm_*_dev on row-major factors (the solve kernels alone; "
                          "algorithmic bytes sizeof(T) * n^2); solve_cm_ms: rflu_getrs_*_dev on column-major factors (+ one transpose of F); "
                          "solve64_ms: rflu_getrs_rm_*_dev with a row-major n x 64 block of right-hand sides",
            "solve_gbs": round(esz * n * n / (solve_ms * 1e-3) / 1e9, 1) if solve_ms else None,
            "nopivot": {"ms": round(1e3 * t_np, 3), "gflops": round(flops / t_np / 1e9, 1),
                        "frac_of_mfma_peak": round(flops / t_np / 1e12 / PEAK_TFLOPS[sfx], 4),
                        "note": "lu!(A, Val(false)) on the same uniform input (step = refill + lu!); residual not meaningful without pivoting"},
            "butterfly": {"ms": round(1e3 * t_bf, 3), "gflops": round(flops / t_bf / 1e9, 1),
                          "frac_of_mfma_peak": round(flops / t_bf / 1e12 / PEAK_TFLOPS[sfx], 4),
                          "butterfly_mul_ms": round(bf_ms, 4),
                          "butterfly_mul_gbs": round(2.0 * esz * n * n / (bf_ms * 1e-3) / 1e9, 1),
                          "butterfly_mul_frac_of_hbm": round(2.0 * esz * n * n / (bf_ms * 1e-3) / 8e12, 4),
                          "solve_backward_error": berr,
                          "note": "step = refill + A <- U'AV (one pass, 2*sizeof(T)*n^2 algorithmic bytes) + lu!(A, Val(false)); "
                                  "backward error ||A x - b|| / (||A|| ||x||) of x = V ((U'AV) \\ (U'b)), src/butterflylu.jl:45-55"},
        }

    # ---- the reference's own boundary: a host array in, the factors back in it (src/lu.jl:116-121 pins the caller's arrays).  Not part
    # of `value` (inputs resident in HBM); quoted so that the cost of the two PCIe crossings is on record.  The way back overlaps the
    # factorization (driver.cpp: getrf_host); the way in cannot (the first update touches every column).
    host_entry = None
    if single and not args.no_extras and n >= 1024 and args.blocksize == 0:
        regenerate(); barrier()
        Ah = np.ascontiguousarray(A.cpu().numpy())   # the same n*n values in the same order: the library reads them column-major
        if True:
            ih = np.empty(n, dtype=np.int64)
            infh = ctypes.c_int64(0)
            keep = Ah.copy()
            h.set_stream(None)
            ts = []
            for _ in range(3):                     # the first call pins the bounce buffers and faults the pages in
                np.copyto(Ah, keep)
                t0 = time.perf_counter()
                h.call(f"rflu_getrf_{sfx}", n, n, ctypes.c_void_p(Ah.ctypes.data), n, ctypes.c_void_p(ih.ctypes.data if pivot else 0),
                       pivot, 0, ctypes.byref(infh))
                ts.append(time.perf_counter() - t0)
            h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            hh = min(ts[1:])
            host_entry = {"host_to_host_ms": round(1e3 * hh, 2), "factor_ms": round(ms_per_step, 3),
                          "pcie_bytes_each_way": esz * n * n,
                          "note": "rflu_getrf_* on a pageable host array (lu! of a host matrix): copy in, factor, factors and ipiv back in "
                                  "the caller's array; best of two warm calls, wall clock around the call.  Pivoted square / tall "
                                  "8192..16384 (Float64 and, since round 6, Float32): the matrix arrives block column by block column WHILE it is factored (driver.cpp: "
                                  "getrf_host_engine, the persistent update engine of csrc/engine.hip), rows leave as they become final"}
            del keep, Ah

    # ---- checks on the last factorization: residual on device (torch as an independent checker) ----
    check = {}
    if not args.no_check and single and n <= 32768:
        regenerate()
        barrier()
        A0 = A.clone()  # column-major memory; A.T is the logical matrix in torch's row-major view
        step()
        barrier()
        LUm = A.T  # logical n x n
        A0m = A0.T
        ipv = ipiv.cpu().numpy()
        if pivot:
            perm = np.arange(n)
            for i, t in enumerate(ipv):
                j = int(t) - 1
                if j != i:
                    perm[i], perm[j] = perm[j], perm[i]
            PA = A0m[torch.from_numpy(perm).to(dev)]
        else:
            PA = A0m
        del A0
        L = torch.tril(LUm, -1)
        L.diagonal().fill_(1)
        U = torch.triu(LUm)
        R = L @ U
        R -= PA
        check["residual_fro"] = float((torch.linalg.norm(R) / torch.linalg.norm(PA)).item())
        check["residual_maxabs"] = float(R.abs().max().item())
        check["info"] = int(info.value)
        del L, U, R, PA

    if not args.no_check and not single:
        if job is not None:
            check["residual_matvec"] = job.matvec_residual()
            check["info"] = int(job.info)
        else:
            regenerate()
            barrier()
            step()
            barrier()
            if rank == 0:
                check["residual_matvec"] = mgpu_matvec_residual(mg, n, slabs, lds, mlayout, args.block, run, pivot)
                check["info"] = int(mg_info[0])

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and single:
        cpu, cpu_ipiv = cpu_baseline(args.cpu_n)
        if pivot and sfx == "f64":
            # ipiv parity on the CPU sample size: same generator, same seed -> must be bit-exact
            m = args.cpu_n
            B = torch.empty((m, m), dtype=tdt, device=dev)
            ip2 = torch.empty(m, dtype=torch.int64, device=dev)
            inf2 = ctypes.c_int64(0)
            h.call(f"rflu_fill_uniform_{sfx}_dev", ctypes.c_void_p(B.data_ptr()), m, m, m, 0, SEED, m, 0, 0, 0.0)
            h.call(f"rflu_getrf_{sfx}_dev", m, m, ctypes.c_void_p(B.data_ptr()), m, ctypes.c_void_p(ip2.data_ptr()), 1,
                   args.blocksize, ctypes.byref(inf2))
            check["ipiv_bit_exact_vs_cpu_oracle_n%d" % m] = bool(np.array_equal(ip2.cpu().numpy(), cpu_ipiv))

    if not single:
        laswp = sweep = None
    if rank == 0:
        out = {
            "metric": "LU GFLOP/s (2n^3/3) on NxN Float64, 1/2/4/8 MI355X; ||PA-LU||/||A||",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            # N > 1 lines are read against one_gpu_same_n (the same n on one GPU in the same run): strong scaling, the form the
            # north star's ">= 6x at 8 GPUs over 1 GPU on N=65536" is stated in; the N = 1 line is the headline configuration
            "scaling": "n/a" if world == 1 else "strong", "vs_baseline": None, "dtype": sfx, "data": "synthetic",
            "config": {"workload": f"lu!(A, ipiv) of a dense uniform[0,1) {n}x{n} {'Float64' if sfx == 'f64' else 'Float32'} "
                                   f"matrix, {'partial pivoting' if pivot else 'NoPivot'}, column-major in HBM",
                       "n": n, "pivot": bool(pivot), "blocksize": args.blocksize,
                       "layout": "single GPU" if world == 1 else
                                 f"1-D block-column cyclic over {world} GPUs (block {args.block}, runs of {run}); " +
                                 ("rflu_getrf_*_mgpu: one process drives all GPUs, ncclBroadcast on the library's streams"
                                  if job is None else "one process per GPU, torch.distributed broadcast (RCCL)"),
                       "timing": "K steps in one bracket (barrier+sync both sides); a step = device refill of the input + lu!"},
            "frac_of_mfma_peak": round(gflops / 1e3 / (PEAK_TFLOPS[sfx] * args.gpus), 4),
            "roofline": roof,
            "laswp": laswp,
            "sweep": sweep,
            "variants": variants,
            "host_entry": host_entry,
            "cpu_baseline": cpu,
            "one_gpu_same_n": one_gpu_same_n,
            "speedup_vs_one_gpu": (round(one_gpu_same_n["ms"] / ms_per_step, 3) if one_gpu_same_n else None),
            "check": check,
            "kernel_ms": {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in kern.items()},
        }
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio (block-buffered when stdout is a pipe): push it out first so
        # that the JSON line is the LAST line of stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
