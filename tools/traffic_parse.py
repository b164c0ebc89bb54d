"""Collect the rocprofv3 --pmc passes of tools/traffic.sh into one JSON: per kernel, HBM bytes per launch
(FETCH_SIZE x 2 on gfx950 for 16-B/lane streaming reads — guides/MI355X_MICROARCH.md, HBM — plus WRITE_SIZE; both counters
are in KiB) and the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 4 SIMDs x 256 CUs ... reported raw)."""
import csv, glob, json, sys, collections, re
out = sys.argv[1]


def instantiation(name):
    """The 3x3 halo kernel runs as several template instantiations with different work per launch (fp16 forward, bf16 data gradient, the
    folded skip convolution, the sub-pixel phase forms): each gets its OWN record, keyed `conv3x3_halo_ws_kernel<f16>`, `<bf16>`, `<f16,+skip>` ...
    (rocprofv3 prints some instantiations demangled - bf16 as "bool _Accum, bool, E" - and some mangled)."""
    base = short(name)
    if not base.startswith("conv3x3_halo") and not base.startswith("conv_phase"):
        return None
    if name.startswith("_Z"):
        m = re.search(r"kernelI(DF16_|DF16b)((?:L[bi]\d+E)*)E", name)
        if not m:
            return None
        typ = "f16" if m.group(1) == "DF16_" else "bf16"
        args = re.findall(r"L([bi])(\d+)E", m.group(2))
        vals = [int(v) for _, v in args]
    else:
        m = re.search(r"kernel<(.*)>\(", name)
        if not m:
            return None
        body = m.group(1)
        typ = "bf16" if "_Accum" in body or "bfloat" in body or "__bf16" in body else "f16"
        vals = [1 if t.strip() == "true" else 0 if t.strip() == "false" else int(t) for t in body.split(",") if t.strip() in ("true", "false") or t.strip().lstrip("-").isdigit()]
    tags = [typ]
    if base == "conv3x3_halo_ws_kernel":          # <T, kPrefetchW, kShape, kFuse, kSkip>
        if len(vals) >= 3 and vals[2]:
            tags.append("+gn")
        if len(vals) >= 4 and vals[3]:
            tags.append("+skip")
    else:
        tags += [str(v) for v in vals]
    return f"{base}<{','.join(tags)}>"


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1)); start = m.end()
        name = name[start:start + n]
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].split("<")[0].strip()


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
