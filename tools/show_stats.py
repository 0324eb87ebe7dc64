import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.2f" % (float(r["TotalDurationNs"]) / 1e6), "%8.3f" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
