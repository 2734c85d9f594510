"""Sum of the named PMC counters per kernel: scripts/probes/pmc_kernel.py <counter_collection.csv> <name-substring> [...]"""
import collections
import csv
import sys

sums = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if not any(p in name for p in sys.argv[2:]):
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
    sums[short][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[short].add(r["Dispatch_Id"])
for k, v in sums.items():
    print(k, "launches", len(launches[k]))
    for c, x in sorted(v.items()):
        print("    %-28s %.4g" % (c, x / len(launches[k])))
