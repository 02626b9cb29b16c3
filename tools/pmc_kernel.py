"""Average of every counter of a rocprofv3 --pmc run per kernel name (substring filter):
python tools/pmc_kernel.py DIR [FILTER]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if flt and flt not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
for k in sorted(acc):
    print(k)
    for cn in sorted(acc[k]):
        s, n = acc[k][cn]
        print("   %-36s %16.1f  (n=%d)" % (cn, s / n, n))
