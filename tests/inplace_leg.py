"""Run as a child process (the pyramid arrangement is chosen once per process from the environment): the front end on device images
with the large-rig pyramid form forced (MORB_PYR_CHAIN=1), so that pyramid level 0 is read IN PLACE in the caller's buffers
(extractor.hip: k_set_l0), through every transition the table has to survive -- tight and padded pitches, a step from host
images in between, a misaligned buffer (copied as before), a camera that keeps its buffer, steps announced ahead -- each
step held against the oracle.  Prints one line `inplace_leg ok ...`; any mismatch raises."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import multi_orb_slam_amd as m
from multi_orb_slam_amd import pipeline, rt, synth
from oracle_pipeline import OracleFrontEnd, assert_same_step

W, H = 640, 480
params = [m.ExtractorParams(nfeatures=1000), m.ExtractorParams(nfeatures=500)]
fe = pipeline.FrontEnd(params, W, H)
ofe = OracleFrontEnd(params, W, H)
T = 12
frames = [[synth.image(c, t, W, H) for c in range(2)] for t in range(T)]
used = []


def dev(img, pitch=None, shift=0):
    """-> (buffer, pointer, pitch): the image in a device buffer with the given row pitch, starting `shift` bytes into it"""
    pitch = pitch or W
    host = np.zeros(H * pitch + shift + 64, np.uint8)
    host[shift:shift + H * pitch].reshape(H, pitch)[:, :W] = img
    b = rt.DeviceBuffer(host.nbytes); b.upload(host)
    return b, b.ptr + shift, pitch


def step_dev(t, pitch=None, shift=(0, 0), gen=None, next_t=None):
    bufs = [dev(frames[t][c], pitch, shift[c]) for c in range(2)]
    rt.device_sync()
    keep.append(bufs)
    args = [(bufs[c][1], bufs[c][2], t + 1) for c in range(2)]
    got = fe.step(args, resident=True)
    assert_same_step(got, ofe.step(frames[t]))
    used.append(fe.ex.level0_in_place())
    return got


keep = []
two = 2
step_dev(0)                                   # tight pitch, aligned: in place
step_dev(1, pitch=W + 64)                     # padded pitch, aligned: in place
assert used == [two, two], used
got = fe.step(frames[2])                      # pageable host images in between: staged by the library (HBM behind the BAR) and read THERE
assert_same_step(got, ofe.step(frames[2])); used.append(fe.ex.level0_in_place())
step_dev(3)                                   # in place again
step_dev(4, shift=(0, 1))                     # camera 1's buffer starts on an odd address: the whole run copies
step_dev(5, pitch=W + 2)                      # a pitch that is no multiple of 4: copies
step_dev(6)
assert used[2:] in ([two, two, 0, 0, two], [0, two, 0, 0, two]), used     # (the staged step: in place where the part has a large BAR)
# steps announced ahead (the extraction of t + 1 .. t + 2 reads ITS buffers in place while step t is matched)
ring = {t: [dev(frames[t][c]) for c in range(2)] for t in range(7, T)}
rt.device_sync()
args_of = lambda t: [(ring[t][c][1], ring[t][c][2], 100 + t) for c in range(2)]
for t in range(7, T):
    nxt = args_of(t + 2) if t + 2 < T else None
    if t == 7:
        fe.announce(args_of(8), resident=True)
    got = fe.step(args_of(t), resident=True, next_images=nxt)
    assert_same_step(got, ofe.step(frames[t]))
fe.close()
n_in = sum(1 for k, u in enumerate(used) if u and k != 2)
print("inplace_leg ok: level 0 read in place in %d of 6 isolated steps on device images, %d overlapped steps" % (n_in, T - 7))
