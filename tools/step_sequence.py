"""Print the dispatch sequence of ONE train step from a rocprofv3 kernel trace (csv): index, kernel, grid, duration.
python tools/step_sequence.py <kernel_trace.csv> [marker kernel that ends a step, default adam]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam"
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1            # the last full step
t0 = int(rows[a]["Start_Timestamp"])
for i, r in enumerate(rows[a:b]):
    name = r["Kernel_Name"]
    for pre in ("(anonymous namespace)::", "_ZN12_GLOBAL__N_1", "void "):
        name = name.replace(pre, "")
    name = name.split("(")[0][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{i:4d} {name:42s} grid {r.get('Grid_Size_X', ''):>8s}x{r.get('Grid_Size_Y', ''):<3s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us")
