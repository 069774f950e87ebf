"""Record the TunableOp results file the s2mel glue ships (voice-tts_amd/tunable/gfx950_s2mel.csv): one pass of the stage at
the bench's production shape with tuning on.  (Only the DiT's shapes: tuning the conditioning encoders' 124 x 512 x 261632
subsampling GEMM made one candidate solution fault the GPU -- do not add exotic shapes without a reason.)  Run on the GPU box:  python tools/tune_gemms.py gpurun_out/gfx950_s2mel.csv"""
import os
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gfx950_s2mel.csv"
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=out, IXTTS_NO_TUNED_GEMMS="1",
                  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="30", PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="5")
sys.argv = [sys.argv[0], "25"]
exec(open(os.path.join(os.path.dirname(__file__), "prof_s2mel.py")).read())
