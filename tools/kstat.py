"""Print calls / average / min / max (us) of the kernels whose name contains one of the given words, from a rocprofv3
*_kernel_stats.csv.  usage: kstat.py stats.csv word [word ...]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(w in n for w in sys.argv[2:]):
        print("%-70s calls %4s  avg %9.1f  min %9.1f  max %9.1f us" % (n.split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
