"""Parse a rocprofv3 --kernel-trace CSV of tools/prof_bigvgan.py: the last forward's dispatches in order (duration, gap to the
previous one) and totals per kernel family.  python tools/bv_trace.py <dir-with-kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X") or r.get("Grid_Size", ""), r.get("Workgroup_Size_X", "")) for r in rows]
idx = [i for i, k in enumerate(ks) if "conv_post_kernel" in k[0]]
lo, hi = idx[-2] + 1, idx[-1] + 1
seq = ks[lo:hi]
tot, gaps, fam = 0, 0, {}
prev = None
for name, s, e, g, w in seq:
    short = name.split("(")[0].replace("void ixtts::", "").replace("ixtts::", "")
    d = (e - s) / 1e3
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    tot += d
    gaps += max(gap, 0)
    fam[short] = fam.get(short, [0, 0.0])
    fam[short][0] += 1
    fam[short][1] += d
    if "-v" in sys.argv:
        print(f"{short[:44]:44s} grid {g:>8s} dur {d:8.1f} us gap {gap:6.1f}")
print(f"forward: {len(seq)} dispatches, kernel time {tot/1e3:.2f} ms, gaps {gaps/1e3:.2f} ms, span {(seq[-1][2]-seq[0][1])/1e6:.2f} ms")
for k, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k[:60]:60s} x{n:4d} {d/1e3:8.2f} ms")
first = seq[0]
print("first dispatch:", first[0][:60], f"{(first[2]-first[1])/1e3:.1f} us")
