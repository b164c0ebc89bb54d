"""Whole-step HBM traffic at one configuration: PMC bytes per launch (profiles/rNN_traffic.json) x launches per step of the serial kernel trace
(profiles/rNN_<cfg>_train_serial_kernel_stats.csv).   python tools/step_traffic.py [cfg2] [r06] [steps in the trace = 63]
The 3x3 halo / sub-pixel kernels are matched per template instantiation (fp16 forward, bf16 data gradient, folded skip convolution ...)."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import instantiation, short
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 63
traffic = json.load(open(f"profiles/{tag}_traffic.json"))[cfg]["kernels"]
calls, usec = {}, {}
for row in csv.DictReader(open(f"profiles/{tag}_{cfg}_train_serial_kernel_stats.csv")):
    k = instantiation(row["Name"])
    if k not in traffic:
        k = short(row["Name"])
    if k not in traffic:
        continue
    calls[k] = calls.get(k, 0) + int(row["Calls"]); usec[k] = usec.get(k, 0) + int(row["TotalDurationNs"]) / 1e3
tot = 0.0
rows = []
for k, n in calls.items():
    gb = traffic[k]["hbm_bytes_per_launch"] * n / steps / 1e9
    tot += gb
    rows.append((gb, k, n / steps, traffic[k]["hbm_bytes_per_launch"] / 1e6, usec[k] / steps / 1e3))
for gb, k, n, mb, ms in sorted(rows, reverse=True):
    print(f"{k:44s} {n:6.1f} launches/step x {mb:8.1f} MB = {gb:6.2f} GB   ({ms:6.2f} ms: {gb / ms * 1e3 if ms else 0:6.0f} GB/s)")
print(f"total {tot:.1f} GB per step")
