"""profiles/gemm_traffic.json from a PMC summary (scripts/pmc_summary.py output, e.g. profiles/r03a_n16384_pmc.txt): mean HBM bytes
per gemm_sub_kernel launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: FETCH_SIZE under-reports reads by 2x on
gfx950), stamped with the hash of the kernel sources it was measured on -- bench.py reports roofline.traffic only while that
hash matches the build it runs.
usage: python scripts/make_gemm_traffic.py profiles/rNN_n16384_pmc.txt 16384 f64"""
import importlib.util, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_rflu_build", os.path.join(ROOT, "recursivefactorization.jl_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
path, n, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals = {}
for line in open(path):
    m = re.match(r"gemm_sub_kernel\s+(FETCH_SIZE|WRITE_SIZE)\s+launches=\s*(\d+)\s+sum=\S+\s+mean=(\S+)", line)
    if m:
        vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
out = {"n": n, "dtype": dtype, "hbm_bytes_per_launch": int((2 * fetch[1] + write[1]) * 1024),
       "sources_sha1": B.sources_digest(),
       "source": f"{os.path.relpath(path, ROOT)}: gemm_sub_kernel mean FETCH_SIZE {fetch[1]:.0f} KiB (x2 gfx950 read correction) + mean "
                 f"WRITE_SIZE {write[1]:.0f} KiB over the {fetch[0]} gemm_sub_kernel launches of the PMC run"}
json.dump(out, open(os.path.join(ROOT, "profiles", "gemm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
