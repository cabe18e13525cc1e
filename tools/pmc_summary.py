"""Aggregate a rocprofv3 --pmc CSV (counter_collection.csv) per kernel: python tools/pmc_summary.py DIR > out.txt"""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per-kernel mean of each counter over its dispatches (rocprofv3 --pmc); n = dispatches")
for k, cs in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = max(len(v) for v in cs.values())
    line = "%-90s n=%-4d " % (k, n) + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        # gfx94x derived-counter formula (rocprofv3 has no gfx950 section): MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * CUs * 4 SIMDs)
        line += "  MfmaUtil=%.1f%%" % (100.0 * busy / (act / 8 * 256 * 4)) if act else ""  # GUI_ACTIVE is summed over the 8 XCDs
    print(line)
