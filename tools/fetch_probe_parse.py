"""FETCH_SIZE / WRITE_SIZE per kernel of tools/fetch_probe.hip against the bytes each kernel is known to read: python tools/fetch_probe_parse.py DIR"""
import csv, glob, sys, collections
GiB = 1 << 30
known = {"reg_full": GiB, "dma_full": GiB - 65536, "reg_half": GiB // 2, "dma_half": (GiB - 65536) // 2, "store_full": GiB}
rows = collections.defaultdict(list)
r_names = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0]
        r_names[(name, int(r["Dispatch_Id"]))] = r["Kernel_Name"]
        rows[(name, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for (name, counter), vals in sorted(rows.items()):
    vals.sort()
    if name == "reg_full":                       # odd dispatches of reg_full are the cache flushes (same byte count)
        pass
    for i, (d, v) in enumerate(vals):
        tag = name
        if name == "reg_half":
            tag = "reg_half" if i % 2 == 0 else "reg_half_twice"
        if name.startswith("void pair_half") or name.startswith("pair_half"):
            full = r_names[(name, d)]
            tag = ("pair_same_xcd_dma aux " + full.split("<")[1].split(">")[0].split(",")[-1].strip() if "true" in full else "pair_same_xcd_reg")      # (launched with start delays 0 / 2 / 4 us, in that order per kind)
        b = known.get(name, 0) or (GiB - 65536) // 2
        print(f"{tag:26s} dispatch {d:3d} {counter:10s} = {v * 1024 / 1e6:10.1f} MB (counter x 1 KiB)   known bytes {b / 1e6:8.1f} MB   counter / known = {v * 1024 / b if b else 0:.3f}")
