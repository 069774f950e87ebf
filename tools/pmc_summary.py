"""Summarise rocprofv3 --pmc passes (CSV output) into the small JSON files kept under profiles/.

  python tools/pmc_summary.py traffic FETCH_DIR WRITE_DIR OUT.json "<command>"    per-kernel HBM bytes per launch (+ per forward)
  python tools/pmc_summary.py mfma DIR OUT.json "<command>"                       per-kernel MFMA-busy share

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly 1/2 of the bytes of a wide
coalesced streaming read (16 B/lane) -> x2; WRITE_SIZE (KiB) reads exactly.  The two do not fit one pass (TCC slots), hence two
runs of the same command.  SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES share one SQ pass.
"""
import collections
import csv
import glob
import json
import statistics
import sys


def load(d):
    """{kernel: {counter: [value per dispatch]}} of one pass directory."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def short(k):
    return k.split("(")[0].replace("void ", "")


def traffic(fetch_dir, write_dir, dst, command):
    fetch, write = load(fetch_dir), load(write_dir)
    res, tot_f, tot_w = {}, 0.0, 0.0
    for k in sorted(fetch):
        f = fetch[k].get("FETCH_SIZE", [])
        w = write.get(k, {}).get("WRITE_SIZE", [0.0])
        if not f:
            continue
        tot_f += sum(f)
        tot_w += sum(w)
        if not k.startswith(("void ixtts", "ixtts")):
            continue
        fm, wm = statistics.median(f), statistics.median(w)
        res[k] = {"launches": len(f), "FETCH_SIZE_KiB_raw_median": fm, "WRITE_SIZE_KiB_median": wm, "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0}
    out = {"command": command, "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 halves FETCH_SIZE on 16 B/lane streams)", "kernels": res}
    n_fwd = sum(len(v.get("FETCH_SIZE", [])) for k, v in fetch.items() if "conv_post_kernel" in k)
    if n_fwd:  # a BigVGAN run: whole-forward traffic = every ixtts dispatch of the run / forwards in it
        fi = sum(sum(v["FETCH_SIZE"]) for k, v in fetch.items() if "ixtts" in k)
        wi = sum(sum(v.get("WRITE_SIZE", [])) for k, v in write.items() if "ixtts" in k)
        out["bigvgan_forward"] = {"forwards": n_fwd, "hbm_bytes_per_forward": (2.0 * fi + wi) * 1024.0 / n_fwd,
                                  "fetch_bytes_per_forward": 2.0 * fi * 1024.0 / n_fwd, "fetch_bytes_per_forward_uncorrected": fi * 1024.0 / n_fwd,
                                  "write_bytes_per_forward": wi * 1024.0 / n_fwd,
                                  "caveat": "the x2 FETCH_SIZE correction is calibrated for 16 B/lane streams (the split-product convs' weight and x-plane "
                                            "loads are such streams); the Snake passes read 4 B/lane (and the fp32-MFMA build's x tiles are 4 B/lane LDS-DMA): "
                                            "for those the true fetch volume lies between the two figures"}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.3f} MB  x{v['launches']:5d}  {short(k)[:100]}")
    if n_fwd:
        print("bigvgan forward:", {k: round(v / 1e6, 1) if isinstance(v, float) else v for k, v in out["bigvgan_forward"].items()})


def mfma(d, dst, command):
    c = load(d)
    res = {}
    for k in sorted(c):
        if not k.startswith(("void ixtts", "ixtts")):
            continue
        m, b = c[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []), c[k].get("SQ_BUSY_CYCLES", [])
        if not m or not b or sum(m) == 0:
            continue
        q = sum(m) / max(sum(b), 1.0)
        res[k] = {"launches": len(m), "SQ_VALU_MFMA_BUSY_CYCLES_sum": sum(m), "SQ_BUSY_CYCLES_sum": sum(b), "mfma_busy_over_sq_busy": q,
                  "mfma_busy_fraction": q / 32.0}
    json.dump({"command": command, "note": "SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 SIMDs, SQ_BUSY_CYCLES the busy cycles of 32 shader-engine "
               "instances (8 XCDs x 4): mfma_busy_fraction = quotient / 32 = share of SIMD-cycles with the matrix pipe busy.  Calibration: the 128x128 conv "
               "tile at 105 TFLOP/s = 0.67 of the fp32 MFMA peak reads 0.67.", "kernels": res}, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{v['mfma_busy_fraction']:6.3f} busy  x{v['launches']:5d}  {short(k)[:100]}")


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    else:
        mfma(*sys.argv[2:5])
