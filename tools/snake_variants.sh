set -e
bash tools/gpu_entry.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ns1 ns2 default ns8; do
  if [ $v = default ]; then unset IXTTS_LIB; else export IXTTS_LIB=$GRAFT_REPO_ROOT/voice-tts_amd/libixtts_hip_$v.so; fi
  IXTTS_BV_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_snk_$v -- python3 tools/prof_bigvgan.py 1892 > gpurun_out/r03/snk_$v.log 2>&1
  f=$(find gpurun_out/r03/prof_snk_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep 'bigvgan F' gpurun_out/r03/snk_$v.log) | snake planes total ns (5 fwd): $(grep aa_snake_planes $f | awk -F'","' '{print $3, $4}')"
  rm -rf gpurun_out/r03/prof_snk_$v
done
