"""Development aid: per-kernel means of a rocprofv3 --pmc run with durations, and rates derived from them
(GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_* over the chip)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in d.items()}
    print(k)
    for n, v in sorted(m.items()):
        print("    %-28s %14.0f" % (n, v))
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print("    clock %.2f GHz" % (cyc / m["dur_ns"]))
        if "SQ_BUSY_CU_CYCLES" in m: print("    CU busy %.2f of the launch" % (m["SQ_BUSY_CU_CYCLES"] / 256 / cyc))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print("    MFMA busy %.2f of the launch" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
        if "SQ_LDS_IDX_ACTIVE" in m: print("    LDS active %.2f, bank conflicts %.2f of the launch" % (m["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, m.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / cyc))
