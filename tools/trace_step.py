"""List the kernels of the last train step of a rocprofv3 kernel trace with start / end offsets (us) and queue."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("(", "<"):
        n = n.split(cut)[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("::")[-1] or "?", r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
a, b = ends[-2], ends[-1]
t0 = rows[a][1]
for s, e, n, q in rows[a + 1:b + 1]:
    if (e - s) > 20000:
        print("%9.1f %9.1f %8.1f q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
