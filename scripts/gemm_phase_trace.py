"""Where does a bulk-GEMM workgroup spend its time, and are the two workgroups of a CU in step?  Needs the stamped build
(scripts/build_gemm_trace.sh, RFLU_LIB=.../librflu_gemmtrace.so): every workgroup of gemm_sub_kernel leaves wall-clock stamps
(100 MHz) at entry, after the first LDS fill, after the K loop, after the C stores were issued / acknowledged, and its HW_ID.
usage: RFLU_LIB=recursivefactorization.jl_amd/librflu_gemmtrace.so python scripts/gemm_phase_trace.py [S] [K] [masked]"""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recursivefactorization.jl_amd import _ffi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
masked = len(sys.argv) > 3 and sys.argv[3] == "masked"
SN = int(os.environ.get("SN", S))      # columns (default: square)
WARM = int(os.environ.get("WARM", "3"))   # launches of the same shape right before the stamped one (clock ramp)
lib = _ffi.load()
h = _ffi.Handle(0)
dev = torch.device("cuda", 0)
if masked: h.set_stream(h.update_stream())
else: h.set_stream(None)
P = lambda t: ctypes.c_void_p(t.data_ptr())
A = torch.rand((S, K), dtype=torch.float64, device=dev)
B = torch.rand((K, SN), dtype=torch.float64, device=dev) * 1e-3
C = torch.rand((S, SN), dtype=torch.float64, device=dev)
nwg = ((S + 127) // 128) * ((SN + 127) // 128)
Aw = torch.rand((15872, 512), dtype=torch.float64, device=dev); Bw = torch.rand((512, 15872), dtype=torch.float64, device=dev) * 1e-3; Cw = torch.rand((15872, 15872), dtype=torch.float64, device=dev)
st = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)
lib.rflu_debug_gemm_stamps.argtypes = [ctypes.c_void_p]; lib.rflu_debug_gemm_stamps.restype = None
h.call("rflu_gemm_rm_f64_dev", S, SN, K, P(A), K, P(B), SN, P(C), SN)
torch.cuda.synchronize()
# warm-up launches of the big shape WITHOUT stamps (pointer still null when they are enqueued), then the stamped launch right behind them
for _ in range(WARM): h.call("rflu_gemm_rm_f64_dev", 15872, 15872, 512, P(Aw), 512, P(Bw), 15872, P(Cw), 15872)
lib.rflu_debug_gemm_stamps(P(st))
h.call("rflu_gemm_rm_f64_dev", S, SN, K, P(A), K, P(B), SN, P(C), SN)
h.synchronize(); torch.cuda.synchronize()
lib.rflu_debug_gemm_stamps(None)
s = st.cpu().numpy()
t00 = s[:, 0].min()
tend = s[:, 5].max()
tick = 0.01  # us
print(f"S={S}x{SN} K={K} {'masked 224 CUs' if masked else 'all CUs'} after {WARM} warm launches: kernel span {(tend - t00) * tick:.1f} us = {2.0 * S * SN * K / ((tend - t00) * tick) / 1e6:.2f} TFLOP/s, {nwg} workgroups")
import numpy as np
pro = (s[:, 1] - s[:, 0]) * tick; main = (s[:, 2] - s[:, 1]) * tick; epi = (s[:, 3] - s[:, 2]) * tick; ack = (s[:, 5] - s[:, 3]) * tick
for name, v in (("prologue (entry -> first LDS fill done)", pro), ("K loop", main), ("C stores issued", epi), ("stores acknowledged", ack)):
    print(f"   {name:42s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p50 {np.percentile(v, 50):7.2f}   p90 {np.percentile(v, 90):7.2f}   max {v.max():7.2f}")
if os.environ.get("SPLIT"):   # RFLU_GEMM_TRACE=2 build: words 6 / 7 are wall-clock stamps inside the prologue
    for name, v in (("entry -> operand slab 0 and C rows requested", (s[:, 6] - s[:, 0]) * tick), ("-> slab 0 arrived and in LDS", (s[:, 7] - s[:, 6]) * tick), ("-> barrier passed, rest of C requested", (s[:, 1] - s[:, 7]) * tick)):
        print(f"   {name:50s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p50 {np.percentile(v, 50):7.2f}   p90 {np.percentile(v, 90):7.2f}")
    s[:, 6] = 0; s[:, 7] = 1
cyc = (s[:, 7] - s[:, 6]).astype(float)
ghz = cyc / (main * 1e3)
ideal = (K / 16) * 64 * 64
print(f"   K loop in shader clocks: mean {cyc.mean():9.0f}  p10 {np.percentile(cyc, 10):9.0f}  p90 {np.percentile(cyc, 90):9.0f}   = {cyc.mean() / ideal:.3f} x the {ideal} clocks of its MFMAs;  shader clock {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f})")
# residency: group by CU
hw = s[:, 4]
xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
cu = (xcc << 16) | ((hwid >> 8) & 0xff) | (((hwid >> 13) & 0x7) << 8)
slot = hwid & 0xf
groups = collections.defaultdict(list)
for i in range(nwg): groups[int(cu[i])].append(i)
print(f"   {len(groups)} distinct (xcc, se, sh, cu) ids; workgroups per id: min {min(len(v) for v in groups.values())} max {max(len(v) for v in groups.values())}; wave slots seen: {sorted(set(int(x) for x in slot))}")
# per CU: time with 0 / 1 / 2 workgroups inside their K loop, between the first entry and the last exit on that CU
tot = [0.0, 0.0, 0.0, 0.0]; span = 0.0
dphase = []
for c, idx in groups.items():
    ev = []
    for i in idx:
        ev.append((s[i, 1], +1)); ev.append((s[i, 2], -1))
    ev.sort()
    lo = min(s[i, 0] for i in idx); hi = max(s[i, 5] for i in idx)
    span += (hi - lo) * tick
    cur = 0; last = lo
    for t, d in ev:
        tot[min(cur, 3)] += (t - last) * tick; last = t; cur += d
    tot[min(cur, 3)] += (hi - last) * tick
    # phase between the two residents: for every workgroup, where inside the K loop of a co-resident does its K loop start?
    for i in idx:
        for j in idx:
            if i != j and s[j, 1] <= s[i, 1] < s[j, 2]:
                dphase.append((s[i, 1] - s[j, 1]) / max(1, (s[j, 2] - s[j, 1])))
print(f"   CU-time by number of resident workgroups inside the K loop: 0: {100 * tot[0] / span:.1f} %   1: {100 * tot[1] / span:.1f} %   2: {100 * tot[2] / span:.1f} %   3+: {100 * tot[3] / span:.1f} %")
if dphase:
    hist, _ = np.histogram(dphase, bins=10, range=(0, 1))
    print("   start of a K loop relative to the co-resident's K loop (deciles of that loop): " + " ".join(str(int(x)) for x in hist))
# timeline of one CU
c0 = sorted(groups)[len(groups) // 2]
print(f"   one CU (id {c0:#x}): slot entry first-fill loop-end stores-acked (us since kernel start)")
for i in sorted(groups[c0], key=lambda i: s[i, 0])[:12]:
    print(f"      wg {i:6d} slot {int(slot[i])}: {(s[i, 0] - t00) * tick:8.2f} {(s[i, 1] - t00) * tick:8.2f} {(s[i, 2] - t00) * tick:8.2f} {(s[i, 5] - t00) * tick:8.2f}")
