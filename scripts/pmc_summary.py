"""Development aid: per-kernel means of the counters in a rocprofv3 --pmc run (counter_collection CSV)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("%-72s %s" % ("kernel (mean per launch)", "  ".join("%22s" % n for n in names)))
for k, d in sorted(acc.items()):
    print("%-72s %s" % (k, "  ".join("%22.0f" % (sum(d[n]) / len(d[n])) if n in d else " " * 22 for n in names)))
