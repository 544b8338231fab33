import csv, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof2'
import os
for f in sorted(glob.glob(d + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(f)):
        us = float(r['AverageNs']) / 1e3
        print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} avg_us={us:8.1f}  {r['Percentage']}%")
