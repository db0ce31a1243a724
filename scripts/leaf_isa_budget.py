"""Instruction budget of ONE pivot step of the cooperative leaf, from the ISA hipcc emits (round-4 review, item 2): the row waves'
straight-line code between two barriers and the communication wave's loop body, by instruction class.
usage: python scripts/leaf_isa_budget.py [PW]      (compiles csrc/panel_local.hip to assembly under /tmp; PW = row waves, default 4)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PW = int(sys.argv[1]) if len(sys.argv) > 1 else 4
asm = "/tmp/panel_local_budget.s"
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result",
                       "--cuda-device-only", "-S", os.path.join(ROOT, "recursivefactorization.jl_amd", "csrc", "panel_local.hip"), "-o", asm],
                      stderr=subprocess.DEVNULL)
s = open(asm).read()
m = re.search(r"^(_ZN4rflu24panel_pivot_local_kernelIdLb0ELi%dELi0EEEvNS_9LocalArgsIT_EE):" % PW, s, re.M)
body = s[m.end():s.index(".Lfunc_end", m.end())]
lines = [l.split(";")[0].strip() for l in body.split("\n")]
lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]

def cls(l):
    op = l.split()[0]
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("ds_"): return "LDS access"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith(("s_load", "s_buffer")): return "scalar memory"
    if "dpp" in l: return "VALU with DPP (cross-lane)"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "ds_bpermute", "ds_swizzle")): return "lane <-> scalar / permute"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64")): return "VALU Float64 arithmetic"
    if op.startswith(("v_div", "v_rcp", "v_ldexp", "v_frexp")): return "VALU division pieces"
    if op.startswith("v_cmp"): return "VALU compare"
    if op.startswith("v_cndmask"): return "VALU select"
    if op.startswith("v_"): return "VALU other (moves, integer, bit ops)"
    if op.startswith(("s_getreg", "s_setprio", "s_sleep", "s_nop", "s_memtime", "s_memrealtime")): return "SALU special"
    if op.startswith("s_"): return "SALU"
    return "other"

bar = [n for n, l in enumerate(lines) if l.startswith("s_barrier")]
def budget(a, b):
    c = collections.Counter(cls(l) for l in lines[a:b])
    return c, b - a
def show(title, a, b):
    c, n = budget(a, b)
    print(f"{title}: {n} instructions")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"    {v:5d}  {k}")
print(f"# panel_pivot_local_kernel<double, false, {PW}, 0> (any placement), {len(lines)} instructions, {len(bar)} barriers")
print("# The communication wave runs a run-time loop (two barriers per column); the row waves run 64 unrolled steps (two barriers per column).")
# communication wave: the loop is the first code after the prologue: its body = the instructions between the first barrier pair pattern that repeats
# (found as the first back edge); the row waves' steps are the long straight-line tail.
back = [n for n, l in enumerate(lines) if l.startswith(("s_cbranch", "s_branch"))]
# row waves: take column 32 of 64: barriers come in pairs A(c), B(c); the unrolled part is the last 2*64 barriers
tail = bar[-128:]
a32, b32, a33 = tail[2 * 32], tail[2 * 32 + 1], tail[2 * 33]
show("row waves, column 32: barrier A -> barrier B (hand-over read, two chain multiply-adds, scale, wave argmax, record)", a32, b32)
show("row waves, column 32: barrier B -> barrier A of column 33 (the rest of elimination 31: 31 multiply-adds per row, row record of the candidate's owner)", b32, a33)
a8, b8, a9 = tail[2 * 8], tail[2 * 8 + 1], tail[2 * 9]
show("row waves, column 8: barrier B -> barrier A of column 9 (55 multiply-adds per row)", b8, a9)
# communication wave: a run-time loop { barrier B(c - 1); combine + publish + pause + poll + reduce + divide + hand over; barrier A(c) } in front of the
# unrolled row-wave code: the last two barriers in front of the unrolled tail are B and A of the loop body (static count: the poll loop's body once)
head_bar = [b for b in bar if b < tail[0]]
if len(head_bar) >= 2:
    show("communication wave, loop body: barrier B -> barrier A (combine the PW wave records, publish the header, pause, ONE poll round for the G headers and the winner's row record, finish the lagging entries, reduce, divide, hand over; the poll loop's body counted once)", head_bar[-2], head_bar[-1])
print("# the five things the algorithm needs per column and row wave: |a| compare (the integer-key argmax: 6 DPP steps + 1 readlane on the fast path; ties add 12 DPP steps), the")
print("# reciprocal (communication wave only: every lane divides for its own header while the reduction runs), one multiply (l = a * 1/pivot), and the multiply-adds of the rank-1")
print("# update: 2 on the chain (A -> B), the other <= 61 per row between B and the next A, next to the exchange.  Everything else in the A -> B segment is control: the 48-byte hand-over")
print("# read (3 ds_read_b128), position bookkeeping (v_cmp / v_cndmask pairs), s_nop padding between DPP steps, exec-mask moves.")
