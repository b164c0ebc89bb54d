"""Summarise a rocprofv3 kernel trace (csv): time per kernel name and per (kernel, grid) shape."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"]
    for pre in ("(anonymous namespace)::", "_ZN12_GLOBAL__N_1"):
        name = name.replace(pre, "")
    name = name.split("(")[0][:34]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    d = by[key]
    d[0] += 1
    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in by.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k[0]:36s} grid {k[1]:>8s}x{k[2]:<3s} n={v[0]:5d} avg {v[1]/v[0]:8.1f} us  total {v[1]/1e3:8.2f} ms  {100*v[1]/tot:5.1f}%")
