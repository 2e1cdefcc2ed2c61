import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import multi_orb_slam_amd as m
import helpers, oracle
fr = helpers.make_frame_arrays([4000] * 8, 1920, 1080, 17)
for nq, th in ((36000, 20.0), (12000, 20.0), (36000, 8.0)):
    q = helpers.make_queries(fr, nq, 31, th=th)
    q["cam"] = q["cam"] % 2 + 3
    matcher = m.Matcher()
    F = matcher.frame(m.FrameData(**fr)); OF = oracle.FrameData(**fr)
    n, mo = matcher.SearchByProjection(F, q)
    on, omo = oracle.search_by_projection_frames(OF, q, 100, True)
    print(nq, th, "n", n, on, "status", matcher.last_resolve(), "diff", int((mo != omo).sum()))
    d = np.flatnonzero(mo != omo)[:10]
    print(" idx", d, "got", mo[d], "want", omo[d])
    on2, omo2 = oracle.search_by_projection_frames(OF, q, 100, False)
    matcher.check_orientation = False
    n2, mo2 = matcher.SearchByProjection(F, q)
    print(" no-ori: n", n2, on2, "diff", int((mo2 != omo2).sum()), matcher.last_resolve())
    F.close(); matcher.close()
