"""profiles/engine_traffic.json from the counters of the replayed engine (scripts/pmc_engine.sh -> engine_pmc.txt + engine_replay.txt):
HBM bytes of ONE engine_kernel launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads by 2x
on gfx950), the MFMA-busy share, and the replay's all-in rate, stamped with the hash of the sources they were measured on -- bench.py
reports roofline.traffic for the resident kernel only while that hash matches the build it runs.
usage: python scripts/make_engine_traffic.py profiles/r06x_engine_pmc.txt profiles/r06x_engine_replay.txt 16384 f64"""
import importlib.util, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_rflu_build", os.path.join(ROOT, "recursivefactorization.jl_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
pmc, replay, n, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
vals = {}
for line in open(pmc):
    m = re.match(r"engine_kernel\s+(\S+)\s+launches=\s*(\d+)\s+sum=\S+\s+mean=(\S+)", line)
    if m: vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
runs = [json.loads(l) for l in open(replay) if l.startswith("{")]
best = min(runs, key=lambda r: r["engine_ms"])
hbm = int((2 * vals["FETCH_SIZE"][1] + vals["WRITE_SIZE"][1]) * 1024)
out = {"n": n, "dtype": dtype, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": best["algorithmic_bytes"],
       "traffic_ratio": round(hbm / best["algorithmic_bytes"], 3), "flops_per_launch": best["flops"],
       "replay_alone_ms": best["engine_ms"], "replay_alone_tflops": best["tflops"],
       "sources_sha1": B.sources_digest(),
       "source": f"{os.path.relpath(pmc, ROOT)}: engine_kernel replayed alone (RFLU_ENGINE_REPLAY=1, scripts/pmc_engine.sh): FETCH_SIZE {vals['FETCH_SIZE'][1]:.0f} KiB "
                 f"(x2 gfx950 read correction) + WRITE_SIZE {vals['WRITE_SIZE'][1]:.0f} KiB"}
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD and sums over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' busy clocks
    out["mfma_busy_cycles"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    out["gui_active_cycles_all_xcds"] = vals["GRBM_GUI_ACTIVE"][1]
    out["mfma_busy_frac"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (1024 * vals["GRBM_GUI_ACTIVE"][1] / 8), 4)
json.dump(out, open(os.path.join(ROOT, "profiles", "engine_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
