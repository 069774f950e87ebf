"""Effective shader clock per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass (MI355X_MICROARCH.md, DVFS note:
clock ~ GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time; reads high on dispatches shorter than ~0.3 ms).  python tools/pmc_clock.py DIR"""
import collections, csv, glob, sys
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        if ns < 200_000:  # short dispatches read high
            continue
        a = acc[name.split("(")[0][-48:]]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for k, (c, ns, n) in sorted(acc.items()):
    print(f"{k:48s} x{n:4d}  avg {ns / n / 1e3:8.1f} us  clock {c / 8 / ns:.3f} GHz")
