cd $GRAFT_REPO_ROOT
for sw in 0 1; do for tpb in 4 8 32 125; do
MORB_MATRIX_SWAP=$sw MORB_MATRIX_TPB=$tpb timeout 600 python - <<'PY'
import os, time, torch, numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth
mt = m.Matcher()
n = 32000
d = torch.from_numpy(synth.descriptors(n, 1)).cuda()
out = torch.empty((n, n), dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(): mt.hamming_matrix_device(d.data_ptr(), n, d.data_ptr(), n, out.data_ptr(), s)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
print("swap", os.environ.get("MORB_MATRIX_SWAP"), "tpb", os.environ.get("MORB_MATRIX_TPB"), "%.1f us  %.2f TB/s" % (us, (2.0 * n * n + 64 * n) / us / 1e6), "checksum", int(out.to(torch.int64).sum()))
PY
done; done
