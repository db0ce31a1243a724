"""Summarise rocprofv3 --pmc CSV output per kernel: sum / mean of each counter over the launches of one factorization."""
import csv, collections, re, sys
short = lambda n: re.sub(r"<.*", "", re.sub(r"\(.*", "", n.replace("void rflu::", "")))
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k:28s} {c:28s} launches={len(v):5d} sum={sum(v):.6g} mean={sum(v)/len(v):.6g}")
