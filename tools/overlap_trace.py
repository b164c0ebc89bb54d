"""How much of a train step's backward pass really overlaps: reads a rocprofv3 --kernel-trace CSV, takes the last complete step (between two
adam_kernel launches) and prints, per queue, the busy time and the time both queues are busy, then per kernel name count / mean duration.
   rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -o ov -- python bench.py --config cfg2 --others 0 --sampler_steps 0 --no_cpu --no_profile --steps 4 --warmup 2
   python tools/overlap_trace.py /tmp/ov"""
import csv, glob, re, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-2], adam[-1]
step = rows[lo + 1:hi + 1]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
print(f"step: {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms")
def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:40]
queues = {}
for r in step:
    queues.setdefault(r.get("Queue_Id", "?"), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0][0], iv[0][1]
    for s, e, *_ in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, iv in queues.items():
    print(f"queue {q}: {len(iv)} kernels, busy {union(iv) / 1e6:.3f} ms, sum of durations {sum(e - s for s, e, _ in iv) / 1e6:.3f} ms")
allv = [x for iv in queues.values() for x in iv]
print(f"any queue busy {union(allv) / 1e6:.3f} ms; sum of all durations {sum(e - s for s, e, _ in allv) / 1e6:.3f} ms")
names = {}
for q, iv in queues.items():
    for s, e, n in iv:
        k = (q, n); c, d = names.get(k, (0, 0)); names[k] = (c + 1, d + e - s)
for (q, n), (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  queue {q} {n:34s} x{c:3d}  {d / 1e6:7.3f} ms  mean {d / c / 1e3:8.1f} us")
