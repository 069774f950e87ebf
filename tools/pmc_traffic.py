"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly 1/2 of
the bytes of a wide coalesced streaming read (16 B/lane) -> x2; WRITE_SIZE (KiB) reads exactly.
The two counters do not fit one pass (TCC slots), hence two runs of the same command.
"""
import collections
import csv
import glob
import json
import statistics
import sys


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    fetch, write, dst = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for k in sorted(fetch):
        if not k.startswith(("void ixtts", "ixtts")):
            continue
        f = statistics.median(fetch[k])
        w = statistics.median(write.get(k, [0.0]))
        res[k] = {"launches": len(fetch[k]), "FETCH_SIZE_KiB_raw_median": f, "WRITE_SIZE_KiB_median": w,
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump({"command": "rocprofv3 --pmc <C> --kernel-trace -- python tools/prof_gpt.py bf16 2 137 30",
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 halves FETCH_SIZE on 16 B/lane streams)",
               "kernels": res}, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{v['hbm_bytes_per_launch']/1e6:9.3f} MB  {k[:110]}")


if __name__ == "__main__":
    main()
