#!/bin/bash
# configs[1] (1k batches of 32x64 tableaux, rational solve) from plain C through the asynchronous ABI, ONE batch per launch
# sequence and 16 lanes -- 16,000 tableaux in flight --: what the engine does when the caller is not bench.py's Python
# lane loop (GPU box).  tools/c_stream_cfg1.sh [lanes [steps [lone]]]  (lone = 1: pipamd_engine_set_lone_batches, +4 %)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from piplib_amd import synth
rows = np.stack([synth.lexmin_batch(2000 + 7919 * i, 1000, 63, 32) for i in range(16)])
rows.astype("<i8").tofile("gpurun_out/cfg1_rows.bin")
PY
for rep in 1 2 3; do GPU_MAX_HW_QUEUES=16 examples/batch_stream gpurun_out/cfg1_rows.bin 16 1000 63 32 ${1:-16} ${2:-2560} 0 ${3:-0}; done
rm -f gpurun_out/cfg1_rows.bin
