"""The headline workload through examples/batch_stream (plain C, one host thread, async ABI), GPU box:
python3 tools/c_stream_rate.py [batches [lanes [steps]]]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from piplib_amd import synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 14
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 56
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    path = os.path.join(d, "rows.bin")
    with open(path, "wb") as f:
        for b in range(nb):  # lanes 0..3 of bench.py: no slow-converging tableau among them
            f.write(np.ascontiguousarray(synth.lexmin_batch(1000 + 7919 * b, 10000, 127, 64), dtype="<i8").tobytes())
    for _ in range(3):
        p = subprocess.run([os.path.join(ROOT, "examples", "batch_stream"), path, str(nb), "10000", "127", "64", str(lanes), str(steps)],
                           capture_output=True)
        print(p.stdout.decode().strip() or p.stderr.decode()[-300:], flush=True)
