"""Sums rocprofv3 --pmc counter_collection.csv files per kernel (short name): python scripts/pmc_kernels.py <filter> <csv> [<csv> ...]"""
import collections
import csv
import re
import sys

flt = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0][:70]
        if flt not in n:
            continue
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[n][r["Counter_Name"]] += 1
for n in acc:
    print(n)
    for c in sorted(acc[n]):
        print("   %-28s %16.0f  (%d dispatches)  %14.1f per dispatch" % (c, acc[n][c], calls[n][c], acc[n][c] / calls[n][c]))
