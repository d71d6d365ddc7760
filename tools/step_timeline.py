"""Per-step kernel summary of a rocprofv3 --kernel-trace CSV: for the LAST step of a run (steps are delimited by the first kernel of
the operator set-up, lattice_potential_kernel) -- busy time per kernel name, device idle time, and the largest idle gaps.
    python tools/step_timeline.py <kernel_trace.csv> [marker-substring]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "lattice_potential_kernel<0>"
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
starts = [i for i, (s, e, n) in enumerate(iv) if marker in n]
# steps: groups of marker hits closer than 5 ms belong together
steps = [starts[0]]
for i in starts[1:]:
    if iv[i][0] - iv[steps[-1]][0] > 20e6:
        steps.append(i)
a, b = (steps[-2], steps[-1]) if len(steps) >= 2 else (steps[-1], len(iv))
sel = iv[a:b]
t0, t1 = sel[0][0], max(e for s, e, n in sel)
print("step of %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(sel)))
busy = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in sel:
    busy[short(n)][0] += 1
    busy[short(n)][1] += (e - s) / 1e6
for k, (c, v) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:22]:
    print("  %-50s %5d launches %8.3f ms" % (k, c, v))
# union of intervals -> idle
end, idle, gaps = sel[0][1], 0.0, []
last = sel[0][2]
for s, e, n in sel[1:]:
    if s > end:
        idle += (s - end) / 1e6
        gaps.append(((s - end) / 1e6, (end - t0) / 1e6, short(last), short(n)))
    if e > end:
        end, last = e, n
print("device idle inside the step: %.2f ms" % idle)
for g in sorted(gaps, reverse=True)[:12]:
    print("  gap %6.3f ms at t=%7.2f ms: after [%s] before [%s]" % g)
