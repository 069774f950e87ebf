#!/bin/bash
# Collect the round's judged artefacts on the GPU box in ONE command: the bench line, rocprofv3 kernel stats of the same
# command, the PMC traffic passes (decode GEMVs narrow + wide, one BigVGAN forward) and the MFMA-busy passes (BigVGAN convs,
# DiT attention, rows GEMM).  Usage (from the repo root on the box): bash tools/profile_bench.sh r02
# Counters are collected in their own runs with --kernel-trace only (no other trace domain beside --pmc).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "[profile] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_under_rocprof.json 2> $O/prof.err
find $O/prof -name "*kernel_trace.csv" -delete
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
echo "[profile] kernel stats done"
pmc() {  # pmc <name> <counters...> -- <program args>
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/$name -- python "$@" > $O/$name.log 2>&1
  find $O/$name -name "*kernel_trace.csv" -delete
}
for C in FETCH_SIZE WRITE_SIZE; do
  pmc gpt_$C $C -- tools/prof_gpt.py bf16 2 137 30
  pmc wide_$C $C -- tools/wide_prof.py 8 137 32
  pmc bv_$C $C -- tools/prof_bigvgan.py 1000
done
echo "[profile] traffic passes done"
pmc mfma_bv SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- tools/prof_bigvgan.py 1892
pmc mfma_attn SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- tools/prof_attn_full.py
pmc mfma_rows SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- tools/prof_rows.py
echo "[profile] mfma passes done"
python tools/pmc_summary.py traffic $O/gpt_FETCH_SIZE $O/gpt_WRITE_SIZE $O/pmc_traffic.json "rocprofv3 --pmc <C> --kernel-trace -- python tools/prof_gpt.py bf16 2 137 30" > $O/pmc_traffic.txt
python tools/pmc_summary.py traffic $O/wide_FETCH_SIZE $O/wide_WRITE_SIZE $O/pmc_traffic_wide.json "rocprofv3 --pmc <C> --kernel-trace -- python tools/wide_prof.py 8 137 32" > $O/pmc_traffic_wide.txt
python tools/pmc_summary.py traffic $O/bv_FETCH_SIZE $O/bv_WRITE_SIZE $O/pmc_traffic_bigvgan.json "rocprofv3 --pmc <C> --kernel-trace -- python tools/prof_bigvgan.py 1000" > $O/pmc_traffic_bigvgan.txt
for KP in "bv prof_bigvgan.py 1892" "attn prof_attn_full.py" "rows prof_rows.py"; do
  set -- $KP
  python tools/pmc_summary.py mfma $O/mfma_$1 $O/pmc_mfma_$1.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python tools/$2 $3" > $O/pmc_mfma_$1.txt
done
rm -rf $O/prof $O/gpt_* $O/wide_* $O/bv_* $O/mfma_*
tail -3 $O/pmc_traffic_bigvgan.txt; cat $O/pmc_mfma_bv.txt | tail -4
cat $O/bench.json
