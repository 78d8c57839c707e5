"""Critical-path view of one train step from a rocprofv3 kernel trace (*_kernel_trace.csv): the last step between two
optimiser kernels is cut out; time is split into "a recurrence kernel is running" and "none is", and for the latter the
kernels that cover it (and the idle gaps) are listed in order.  usage: timeline.py trace.csv [out.txt]"""
import csv, sys, collections

def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("(", "<"):
        n = n.split(cut)[0]
    return n.split("::")[-1] or "?"

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam" in r[2]]
a, b = ends[-2], ends[-1]
step = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("step: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)), file=out)
REC = ("persistent", "wide_kernel", "bwd_ps_kernel")
gru = [(s, e) for s, e, n, q in step if any(k in n for k in REC)]
print("recurrence kernels: %d, %.3f ms" % (len(gru), sum(e - s for s, e in gru) / 1e6), file=out)
# segments outside the recurrence kernels
edges = [t0] + [x for s, e in gru for x in (s, e)] + [t1]
segs = [(edges[i], edges[i + 1]) for i in range(0, len(edges), 2)]
tot = collections.Counter()
for k, (s0, s1) in enumerate(segs):
    print("\n-- segment %d: %.3f ms" % (k, (s1 - s0) / 1e6), file=out)
    cur = s0
    inside = [r for r in step if r[1] > s0 and r[0] < s1 and not any(k in r[2] for k in REC)]
    for s, e, n, q in inside:
        s_, e_ = max(s, s0), min(e, s1)
        if s_ > cur:
            print("   idle %7.1f us" % ((s_ - cur) / 1e3), file=out); tot["(idle)"] += s_ - cur
        if e_ > cur:
            tot[n] += e_ - max(cur, s_)
        print("   %-44s q%-3s %8.1f us%s" % (n, q, (e - s) / 1e3, "" if s >= cur else "  (overlaps)"), file=out)
        cur = max(cur, e_)
    if s1 > cur:
        print("   idle %7.1f us" % ((s1 - cur) / 1e3), file=out); tot["(idle)"] += s1 - cur
print("\nexposed time outside the recurrence kernels, by kernel (ms):", file=out)
for n, v in tot.most_common():
    print("   %-44s %7.3f" % (n, v / 1e6), file=out)
