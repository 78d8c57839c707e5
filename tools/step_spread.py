"""Step-to-step spread from a rocprofv3 kernel trace (*_kernel_trace.csv): for every train step (cut at the optimiser kernel) its
length, the eight recurrence launches and the nine segments between them.  usage: step_spread.py trace.csv [out.txt]"""
import csv, sys

def short(n):
    n = n.replace("void ", "")
    for cut in ("(", "<"):
        n = n.split(cut)[0]
    return n.split("::")[-1] or "?"

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
REC = ("persistent", "wide_kernel", "bwd_ps_kernel")
ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
for a, b in zip(ends[:-1], ends[1:]):
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    gru = [(s, e) for s, e, n in step if any(k in n for k in REC)]
    edges = [t0] + [x for s, e in gru for x in (s, e)] + [t1]
    segs = [(edges[i + 1] - edges[i]) / 1e3 for i in range(0, len(edges), 2)]
    print("step %7.3f ms | rec %s | seg %s" % ((t1 - t0) / 1e6, " ".join("%4.0f" % ((e - s) / 1e3) for s, e in gru),
                                              " ".join("%4.0f" % x for x in segs)), file=out)
