cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_matcher.py -x -q -m gpu -k "top2 or matrix" 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 python - <<'PY'
import os, time, torch, numpy as np, ctypes as C
import multi_orb_slam_amd as m
from multi_orb_slam_amd import synth, _lib
mt = m.Matcher()
for n in (4000, 8192, 32000):
    d = torch.from_numpy(synth.descriptors(n, 1)).cuda()
    q = torch.from_numpy(synth.perturbed_queries(synth.descriptors(n, 1), 3)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    bi = torch.empty(n, dtype=torch.int32, device="cuda"); bd = torch.empty_like(bi); sd = torch.empty_like(bi)
    for on in (1, 0):
        m.Matcher.use_matrix_cores(on)
        nb = _lib.lib().orbm_top2_scratch_bytes(n, n)
        scr = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        def run_t(): _lib.lib().orbm_hamming_top2_device(C.c_void_p(q.data_ptr()), n, C.c_void_p(d.data_ptr()), n, C.c_void_p(bi.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_void_p(scr.data_ptr()), C.c_void_p(s))
        for _ in range(3): run_t()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): run_t()
        e1.record(); torch.cuda.synchronize()
        print("n", n, "matrix cores", on, "%.1f us" % (e0.elapsed_time(e1) / 40 * 1e3), "scratch", nb, "checksums", int(bi.to(torch.int64).sum()), int(bd.sum()), int(sd.sum()))
PY
