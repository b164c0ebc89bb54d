"""Sum one rocprofv3 --pmc counter per kernel name:  python tools/pmc_sum.py <dir> [COUNTER=FETCH_SIZE] [name filter]  -> launches, mean value, x 2 KiB for FETCH_SIZE (gfx950)"""
import csv, glob, re, sys
d, ctr = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE")
flt = sys.argv[3] if len(sys.argv) > 3 else ""
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        k = (r["Dispatch_Id"], r["Kernel_Name"])
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    for (did, name), v in per.items():
        m = re.search(r"(\w+_kernel)", name); n = m.group(1) if m else name[:40]
        if flt in n: acc.setdefault(n, []).append(v)
for n, v in acc.items():
    mean = sum(v) / len(v)
    print(f"{n:36s} launches {len(v):4d}  mean {ctr} {mean:14.1f}" + (f"  = {mean * 2048 / 1e6:9.1f} MB (x 2 KiB)" if ctr == "FETCH_SIZE" else ""))
