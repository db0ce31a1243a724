"""Copy the summaries scripts/collect_profiles.sh and the size/microbench runs left under gpurun_out/ into profiles/ with headers
that say what they are and which build (hash of the kernel sources) they belong to.
usage: python scripts/install_profiles.py r04c   (reads gpurun_out/prof_<tag>{,_n4096}/ and gpurun_out/<first three letters of the tag>/)"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_rflu_build", os.path.join(ROOT, "recursivefactorization.jl_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
tag = sys.argv[1]
rnd = tag[:3]
sha = B.sources_digest()[:8]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
rd = lambda *p: open(os.path.join(G, *p)).read()

def stats_header(n):
    return (f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}, profiles/gemm_traffic.json) -- rocprofv3 --kernel-trace --stats -- python bench.py --size {n} --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extras\n"
            "# (3 timed factorizations incl. warm-up in the shipped schedule + 1 profiled single-stream pass; Float64, one MI355X; scripts/collect_profiles.sh)\n"
            "# first table: all kernels of the run (scripts/rocpd_summary.py); second: one timed factorization split by HIP queue\n"
            "# (scripts/rocpd_queues.py): the caller's stream = critical path; the 224-CU stream (updates of the lookahead part, then -- from the\n"
            "# first panel of <= 8192 rows on -- the side stream of the leaf-wise schedule); the 192-CU stream (the updates from the last lookahead\n"
            "# block column on).  gate_wait_kernel time is waiting, not work; laswp_kernel on the caller's stream includes the folded gate wait.\n"
            "# third: the profiled single-stream pass: bench.py's roofline.avg_launch_ms is the gemm_sub_kernel average of THAT pass.\n")

def pmc_header(n):
    return (f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}) -- PMC passes, each its own run with --kernel-trace only (scripts/collect_profiles.sh):\n"
            f"#   rocprofv3 --kernel-trace --pmc <COUNTERS> -f csv -- python bench.py --size {n} --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-extras\n"
            "# Counter collection serialises kernels across queues, so these runs use the block-column lookahead schedule with event edges only\n"
            "# (device-side gates cannot make progress when one kernel runs at a time) -- same kernels, same shapes for the bulk GEMM.  Two\n"
            "# factorizations per run (timed schedule + profiled single-stream pass).  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports\n"
            "# 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> read bytes = 2*FETCH_SIZE*1024.\n")

for n, d in ((16384, f"prof_{tag}"), (4096, f"prof_{tag}_n4096")):
    if not os.path.isdir(os.path.join(G, d)):
        continue
    open(os.path.join(P, f"{tag}_n{n}_kernel_stats.txt"), "w").write(stats_header(n) + rd(d, "kernel_stats.txt"))
    open(os.path.join(P, f"{tag}_n{n}_pmc.txt"), "w").write(pmc_header(n) + rd(d, "pmc.txt"))
    if n == 16384:
        open(os.path.join(P, f"{tag}_n{n}_blocks.txt"), "w").write(
            f"# round {int(rnd[1:])}, shipped build -- same rocprofv3 --kernel-trace run as {tag}_n{n}_kernel_stats.txt, one timed factorization\n"
            "# (scripts/rocpd_blocks.py): when every block column's first leaf starts on the critical-path queue, and how busy each queue is\n"
            "# per 5 ms window.  Queue 1 = caller's stream (chain); queue 3 = the 224-CU stream: update stream of the lookahead part, side stream of the\n"
            "# leaf-wise part; queue 4 = the 192-CU stream: the updates from the last lookahead block column on (round 4, DESIGN.md section 3.11).\n"
            + rd(d, "blocks.txt"))

# ---- round 6: the chain leaf by leaf, the engine's counters (replayed alone) and its workgroup-time table
d16 = f"prof_{tag}"
if os.path.exists(os.path.join(G, d16, "leaves.txt")):
    open(os.path.join(P, f"{tag}_n16384_leaves.txt"), "w").write(
        f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}) -- same rocprofv3 --kernel-trace run as {tag}_n16384_kernel_stats.txt, one timed factorization\n"
        "# (scripts/rocpd_leaves.py): the critical-path queue leaf by leaf.  `lookahead launch` = leaf_la_kernel: ~15-21 us of work (interchanges on the next\n"
        "# leaf's 64 columns, diagonal inverse, 64-row solve) + the folded wait for the update engine's progress word.  The waiting is at the LAST leaf of every\n"
        "# block column (g = 8 b + 7): its lookahead strip is the first 64 columns of the NEXT block column, whose column block has to have received BIG(b - 1)\n"
        "# and then, one two-stage leaf window after the other, the seven leaves of block column b factored meanwhile (DESIGN.md section 7).\n"
        + rd(d16, "leaves.txt"))
if os.path.exists(os.path.join(G, d16, "engine_pmc.txt")):
    open(os.path.join(P, f"{tag}_engine_pmc.txt"), "w").write(
        f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}) -- the resident engine_kernel REPLAYED ALONE (RFLU_ENGINE_REPLAY=1, scripts/engine_replay.py: a real\n"
        "# factorization first, then the engine on that image with every leaf counted as done and no chain next to it; same operations, addresses and counts),\n"
        "# three separate passes: rocprofv3 --kernel-trace --pmc <COUNTERS> -f csv -- python scripts/engine_replay.py 16384 1   (scripts/pmc_engine.sh).\n"
        "# ONE launch per run.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> read bytes = 2*FETCH_SIZE*1024.\n"
        "# SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: MFMA-busy share = busy / (1024 * active / 8).\n"
        + rd(d16, "engine_pmc.txt") + "# kernel durations in the three counter runs (rocpd):\n" + "".join("# " + l for l in open(os.path.join(G, d16, "engine_pmc_durations.txt"))))
    open(os.path.join(P, f"{tag}_engine_replay.txt"), "w").write(
        f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}) -- python scripts/engine_replay.py 16384 3 (no profiler): the engine alone on 224 CUs, all-in:\n"
        "# flops = sum of 2 M N K over its operations, algorithmic_bytes = per operation A and B once, the Schur block in and out, the solved block row in and\n"
        "# out, the interchanges (driver.cpp: factor_leafwise), engine_ms = HIP event pair around the one launch.  NOTE the replay has a chain of its own: 256 leaf\n"
        "# windows per column block, two dependent stages each, ~0.2 ms per window -- it is not a pure throughput figure (DESIGN.md section 7).\n"
        + "".join(l for l in open(os.path.join(G, d16, "engine_replay.txt")) if "amdgpu.ids" not in l))
if os.path.exists(os.path.join(G, rnd, "engine_trace.txt")):
    open(os.path.join(P, f"{tag}_engine_workgroup_time.txt"), "w").write(
        f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}) -- RFLU_ENGINE_TRACE=72:24 python scripts/time_env.py 16384 3 (the accounting instantiation of the kernel):\n"
        "# the 448 workgroups' time by kind of unit (100 MHz clock, summed over the workgroups); the scheduler's scan phase by phase, microseconds per call; for the leaves\n"
        "# 72..95 (block columns 9-11) when LEAF(g) was first claimed on the column block of its first columns and -- behind `||` -- on the NEXT block column's first column block\n"
        "# (the catch-up behind BIG(b - 1) the chain's last leaf waits for); and behind every block column's last leaf what stands between it and BIG(b) being complete on the\n"
        "# column block the chain needs next: the block column's own deferred interchanges, the 512-row block solve (stage 0), the tiles.\n"
        + "".join(l for l in open(os.path.join(G, rnd, "engine_trace.txt")) if "amdgpu.ids" not in l))

# ---- size table
rows = []
for f in sorted(os.listdir(os.path.join(G, rnd))):
    if f.startswith("bench_n") and f.endswith(".json"):
        d = json.loads(rd(rnd, f))
        c = d.get("check") or {}
        res = c.get("residual_fro", "n/a (n > 32768: checked by tests/test_gpu_configs.py)")
        rows.append((d["dtype"], not d["config"].get("pivot", True), d["config"]["n"],
                     f"{d['config']['n']:6d}  {d['dtype']}   {str(d['config'].get('pivot', True)):5s} {d['ms_per_step']:10.3f} {d['value'] / 1e3:9.2f}  {d['frac_of_mfma_peak']:.4f}   {res}"))
rows.sort()
dflt = json.loads(rd(rnd, "bench_default.json"))
out = [f"# round {int(rnd[1:])}, shipped build (sources sha1 {sha}): python bench.py --size N [--dtype f32] [--nopivot] --steps 5|3 --warmup 1 --no-cpu-baseline --no-extras",
       "# one MI355X box, one gpurun call (boxes of the pool differ by 3-5 % on the GEMM-bound sizes); the JSON lines are under gpurun_out/" + rnd + "/ (scratch)",
       "# n      dtype pivot   ms/step   TFLOP/s  frac of dense MFMA peak (78.6 f64 / 157.3 f32)   ||PA-LU||/||A||"]
out += [r[3] for r in rows]
out += ["", "# default bench line of the same call (python bench.py --steps 8 --warmup 2), extra keys:"]
rf = dflt["roofline"]
out.append("# roofline: " + json.dumps({k: rf.get(k) for k in ("achieved", "frac", "achieved_profiled", "frac_profiled", "achieved_in_schedule", "frac_in_schedule", "avg_launch_ms", "traffic") if k in rf}))
out.append("# host_entry: " + json.dumps(dflt.get("host_entry")))
lw = dflt.get("laswp") or {}
out.append(f"# laswp.wide: {json.dumps(lw.get('wide'))}  all launches: {lw.get('achieved')} GB/s, {lw.get('total_ms')} ms")
out.append(f"# laswp.alone: {json.dumps(lw.get('alone'))}")
out.append("# sweep: " + json.dumps(dflt.get("sweep")))
out.append("# variants: " + json.dumps(dflt.get("variants")))
cb = dflt.get("cpu_baseline") or {}
out.append("# cpu_baseline: " + json.dumps({k: cb.get(k) for k in ("value", "unit", "cores", "kind")}))
out.append(f"# ms_per_step {dflt['ms_per_step']}  value {dflt['value']}  check {json.dumps(dflt.get('check'))}")
def block(title, *files, keep=""):
    out.append("")
    out.append(title)
    for f in files:
        p = os.path.join(G, rnd, f)
        if os.path.exists(p):
            out.extend("# " + l.rstrip() for l in open(p) if l.strip() and "amdgpu.ids" not in l and keep in l)
block("# bulk GEMM alone (scripts/microbench_gemm_sustained.py: 15872^2 x 512 back to back, f64 then f32; scripts/microbench_gemm_k.py: 15360^2 x K):",
      "gemm_sustained.txt", "gemm_sustained_f32.txt", "gemm_k.txt")
block("# row interchanges alone (scripts/microbench_laswp.py: 512 interchanges x 16384 columns, C-ABI call = bookkeeping kernel + laswp_kernel):", "laswp_alone.txt")
block("# host-pointer entry (scripts/microbench_host_entry.py, one caller buffer refilled in place):", "host_entry.txt")
block("# cooperative leaf alone on the GPU (scripts/panel_bench.py, mode 2 = shipped kernel):", "panel_bench.txt", keep="mode 2")
block("# solve step ldiv!(F, B) on row-major device factors (scripts/microbench_getrs.py):", "getrs.txt")
block("# blocks of right-hand sides: the cooperative MFMA chain against the recursive splitting (scripts/getrs_check.py):", "getrs_block.txt")
block("# the persistent update engine against the stream schedule and its own variants (scripts/time_env.py: alternating in one process, best / median of 4, ms):", "engine_time.txt")
block("# cooperative leaf alone: any placement (mode 2, shipped routing above 4096 rows) against XCD-local (mode 1) (scripts/panel_bench.py):", "panel_bench_local.txt")
block("# one block column of a tall matrix by itself = the owner's panel of the multi-GPU driver (scripts/tall_panel.py):", "tall_panel.txt")
block("# liveness (scripts/engine_stress.py):", "engine_stress.txt")
open(os.path.join(P, f"{tag}_sizes.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
