import csv, glob, sys, collections, os
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(root + '/pmc_*')):
    fs = sorted(glob.glob(d + '/**/*_counter_collection.csv', recursive=True), key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].replace('void wgs::', '')[:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        v2 = v[len(v)//2:]  # skip warmup half
        print(f"   {c:24s} avg={sum(v2)/len(v2):14.1f} n={len(v2)}")
