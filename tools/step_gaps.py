"""From a rocprofv3 kernel trace of tools/prof_gpt.py: per decode step, time in kernels vs time between kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# decode region = from the first sampler_kernel to the end
i0 = next(i for i, e in enumerate(ev) if "sampler_kernel" in e[2])
ev = ev[i0:]
steps = sum("sampler_kernel" in e[2] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
import statistics
print(f"steps {steps} kernels/step {len(ev)/steps:.1f}  span/step {span/steps/1e3:.1f} us  busy/step {busy/steps/1e3:.1f} us  gaps/step {sum(gaps)/steps/1e3:.1f} us  median gap {statistics.median(gaps)/1e3:.2f} us")
from collections import defaultdict
after = defaultdict(list)
for i in range(len(ev) - 1):
    after[ev[i][2][:60]].append(gaps[i])
for k, v in sorted(after.items(), key=lambda kv: -len(kv[1]))[:8]:
    print(f"   gap after {k:60s} n={len(v):5d} median {statistics.median(v)/1e3:.2f} us")
