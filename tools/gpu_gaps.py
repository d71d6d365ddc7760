"""Idle gaps of the device in a rocprofv3 --kernel-trace CSV: every gap above a threshold with the kernels on either side.
    python tools/gpu_gaps.py <kernel_trace.csv> [min_gap_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
end, last = iv[0][1], iv[0][2]
t0 = iv[0][0]
for s, e, n in iv[1:]:
    if s - end > thr * 1e6:
        print("gap %6.1f ms at t=%8.1f ms: after [%s] before [%s]" % ((s - end) / 1e6, (end - t0) / 1e6, short(last), short(n)))
    if e > end:
        end, last = e, n
