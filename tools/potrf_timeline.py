"""Kernel timeline of one geobo_potrf_inv call from a rocprofv3 --kernel-trace CSV (last of the calls in the trace):
    rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python tools/run_potrf_once.py 8448 ctx
    python tools/potrf_timeline.py $(find /tmp/pt -name '*kernel_trace.csv') out.txt"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "potf2" in r["Kernel_Name"]]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 66
first = idx[-nsteps]
t0 = int(rows[first]["Start_Timestamp"])
out = open(sys.argv[2], "w")
R = []
for r in rows[first:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
    R.append((s, e, r.get("Queue_Id", "?"), n))
    out.write("%9.1f %9.1f %7.1f q%s %s\n" % (s, e, e - s, r.get("Queue_Id", "?"), n))
pot = [r for r in R if "potf2" in r[3]]
print("last potf2 ends at %.1f us, everything ends at %.1f us" % (pot[-1][1], max(r[1] for r in R)))
for phase, sel in (("factor", [r for r in R if r[0] < pot[-1][1]]), ("inverse", [r for r in R if r[0] >= pot[-1][1]])):
    d = collections.defaultdict(float)
    for r in sel:
        d[(r[2], r[3][:34])] += r[1] - r[0]
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
        print(phase, k, "%.2f ms" % (v / 1e3))
