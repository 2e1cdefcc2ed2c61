"""One process, a few launches of the 32k x 32k distance matrix (profiling target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
mt = m.Matcher()
d = torch.from_numpy(synth.descriptors(n, 1)).cuda()
out = torch.empty((n, n), dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    mt.hamming_matrix_device(d.data_ptr(), n, d.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print("done", flush=True)
