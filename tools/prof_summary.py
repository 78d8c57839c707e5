"""Condense rocprofv3 csv output (kernel stats or pmc counter collection) into a small summary for profiles/."""
import csv, sys, collections, json

def stats(path, out):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:40]:
            name = r["Name"].split("(")[0].replace("void ", "")
            w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

def pmc(path, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        c = agg[name][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"]); c[1] += 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Sum", "MeanPerDispatch"])
        for k in sorted(agg):
            for cn, (sm, n) in agg[k].items():
                w.writerow([k, cn, n, "%.1f" % sm, "%.1f" % (sm / n)])

if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
