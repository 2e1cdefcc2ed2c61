import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, ctypes
import multi_orb_slam_amd as m, helpers, oracle
from multi_orb_slam_amd import _lib
nq, th = int(sys.argv[1]), float(sys.argv[2])
fr = helpers.make_frame_arrays([1000, 1000], 640, 480, 2)
q = helpers.make_queries(fr, nq, 42, th=th, blocks=1)
mt = m.Matcher(0.8, True)
F = mt.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
n, mo = mt.SearchByProjection(F, q)
st = (ctypes.c_int * 4)(); _lib.lib().orbm_debug_last_resolve(mt._h, st)
on, omo = oracle.search_by_projection_frames(OF, q, 100, True)
print("nq", nq, "th", th, "status", list(st), "ok", n == on and np.array_equal(mo, omo))
