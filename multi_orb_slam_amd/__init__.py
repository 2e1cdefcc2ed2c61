"""multi_orb_slam_amd -- MI355X-native ORB front end (ORBextractor + ORBmatcher hot path of Multi_ORB_SLAM).

The product is the C-ABI shared library `lib/libmorb.so` (hand-written HIP kernels for gfx950, declared in
include/orbx.h, orbm.h, orbf.h and orbv.h) plus the C++ host classes in `host/` that keep the reference's
ORB_SLAM2::ORBextractor / ORBmatcher signatures.  This Python package is a thin ctypes mirror of that ABI used by
the tests, the benchmark and the multi-GPU driver; it contains no compute and no CPU fallback.
"""
from ._lib import lib, build, LIB_PATH, OrbError, KP_DTYPE, QUERY_DTYPE  # noqa: F401
from .extractor import Extractor, ExtractorParams, tables  # noqa: F401
from .matcher import Matcher, FrameData, descriptor_distance, three_maxima  # noqa: F401
from .vocabulary import Vocabulary, BowSearch, Side as BowSide, FeatureVector, score_l1  # noqa: F401
