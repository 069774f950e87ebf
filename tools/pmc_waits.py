"""Per-kernel wave-cycle breakdown from one rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES): python tools/pmc_waits.py DIR"""
import collections, csv, glob, sys
c = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(c.items()):
    if "ixtts" not in k or not v.get("SQ_WAVE_CYCLES"):
        continue
    w = v["SQ_WAVE_CYCLES"]
    print(f"{k.split('(')[0][-52:]:52s} wait_any {v['SQ_WAIT_ANY']/w:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/w:.2f} active {v['SQ_ACTIVE_INST_ANY']/w:.2f} "
          f"wait_lds {v['SQ_WAIT_INST_LDS']/w:.2f} bankconf/wavecyc {v['SQ_LDS_BANK_CONFLICT']/w:.3f} mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/max(v['SQ_BUSY_CYCLES'],1)/32:.2f}")
