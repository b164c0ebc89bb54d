import csv, glob, collections, sys
out = sys.argv[1]; pat = sys.argv[2]
res = {}
for f in sorted(glob.glob(f'{out}/g*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = v[-1]
for k in sorted(res):
    print(f"{k:34s} {res[k]:.5g}")
