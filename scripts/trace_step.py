"""Development aid: from a rocprofv3 kernel trace CSV of bench.py, list the kernels of one replayed training step
(short name, duration in us), and the sums per kernel family."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts with its quantiser launch: the synthesis (adaptive layers), the |w| statistics (TWN / TTQ) or the effective
# weights (INQ); optional second argument: that kernel's name
first = sys.argv[2] if len(sys.argv) > 2 else "synth_fwd_kernel"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
pairs = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if b - a > 50]
a, b = pairs[-4]
seg = rows[a:b]
wall = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
fam = collections.OrderedDict()
for r in seg:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void ", "").replace("aq::", "")
    if "at::native" in name:
        short = "torch:" + name.split("at::native::")[1][:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fam.setdefault(short, [0, 0.0])
    fam[short][0] += 1
    fam[short][1] += d
print("%d kernels, wall %.1f us" % (len(seg), wall))
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %3d  %8.1f us  avg %6.2f" % (k[:70], n, d, d / n))
