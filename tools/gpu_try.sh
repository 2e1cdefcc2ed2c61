cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matcher.py -x -q -m gpu -k "matrix" 2>&1 | grep -E "passed|failed|Error|assert" | head
for n in 32000 8192 4000; do
MATN=$n timeout 600 python - <<'PY'
import os, time, torch, numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
mt = m.Matcher()
n = int(os.environ["MATN"])
d = torch.from_numpy(synth.descriptors(n, 1)).cuda()
out = torch.empty((n, n), dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(): mt.hamming_matrix_device(d.data_ptr(), n, d.data_ptr(), n, out.data_ptr(), s)
for on in (1, 0):
    m.Matcher.use_matrix_cores(on)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(80): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 80 * 1e3
    print("n", n, "matrix cores", on, "%.1f us  %.2f TB/s" % (us, (2.0 * n * n + 64 * n) / us / 1e6))
PY
done
