"""Collect the rocprofv3 --pmc passes of tools/traffic.sh into one JSON: per kernel, HBM bytes per launch
(FETCH_SIZE x 2 on gfx950 for 16-B/lane streaming reads — guides/MI355X_MICROARCH.md, HBM — plus WRITE_SIZE; both counters
are in KiB) and the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 4 SIMDs x 256 CUs ... reported raw)."""
import csv, glob, json, os, sys, collections, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]


from kernel_names import instantiation, short


agg = collections.defaultdict(lambda: collections.defaultdict(list))
clk = collections.defaultdict(lambda: [0.0, 0.0])          # kernel -> [GRBM_GUI_ACTIVE cycles (sum over the 8 XCDs), dispatch ns]
for f in sorted(glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        inst = instantiation(r["Kernel_Name"])
        keys = [short(r["Kernel_Name"])]
        if inst:
            keys.append(inst)
            if inst.endswith("+skip>"):      # the bench line's name for the folded launches; the plain name keeps the plain instantiations only
                keys[0] = keys[0] + "[+1x1 skip]"
        for k in keys:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "End_Timestamp" in r:
                clk[k][0] += float(r["Counter_Value"])
                clk[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, c in agg.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    n = len(c["FETCH_SIZE"])
    fetch = sum(c["FETCH_SIZE"]) / n * 1024 * 2
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    e = {"launches": n, "fetch_bytes_per_launch_corrected": int(fetch), "write_bytes_per_launch": int(write),
         "hbm_bytes_per_launch": int(fetch + write)}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and sum(c["SQ_BUSY_CYCLES"]) > 0:
        # MFMA busy cycles are summed over SIMDs, SQ_BUSY_CYCLES over SEs/XCDs: report the ratio normalised per SIMD
        e["mfma_busy_cycles"] = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        e["gui_active"] = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(c.get("GRBM_GUI_ACTIVE", [0])))
        if e["gui_active"] > 0:
            e["mfma_busy_frac"] = round(e["mfma_busy_cycles"] / (e["gui_active"] / 8 * 1024), 3)   # 8 XCDs summed; 256 CUs x 4 SIMDs
        if clk[k][1] > 0:
            # shader clock while the kernel ran: active cycles per XCD / dispatch time (the counter window is a few us longer than the
            # dispatch, so this overstates short kernels; meaningful for launches of >= 100 us)
            e["sclk_ghz_est"] = round(clk[k][0] / 8 / clk[k][1], 3)
            e["avg_dispatch_us_under_pmc"] = round(clk[k][1] / max(1, len(c.get("GRBM_GUI_ACTIVE", [0]))) / 1e3, 1)
    res[k] = e
print(json.dumps(res, indent=1))
