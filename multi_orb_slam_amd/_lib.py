"""ctypes loader of lib/libmorb.so.  Fails loudly when the HIP library is missing: there is no fallback path."""
import ctypes as C
import os
import subprocess
import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
# MORB_LIB_PATH: load an instrumented build of the same library (csrc/Makefile PHASES=1) instead; experiments only
LIB_PATH = os.environ.get("MORB_LIB_PATH") or os.path.join(_PKG, "lib", "libmorb.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("ur", "<f4"), ("min_level", "<i4"),
                        ("max_level", "<i4"), ("cam", "<i4"), ("blocks", "<i4"), ("angle", "<f4"),
                        ("desc", "u1", (32,))])
WINDOW_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("cam", "<i4"), ("min_level", "<i4"), ("max_level", "<i4")])
assert KP_DTYPE.itemsize == 28 and QUERY_DTYPE.itemsize == 68 and WINDOW_DTYPE.itemsize == 24

ORB_OK, ORB_E_ARG, ORB_E_HIP, ORB_E_CAPACITY, ORB_E_NO_DEVICE, ORB_E_TIMEOUT = 0, -1, -2, -3, -4, -5


class OrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmorb error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):  # orbx_params
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class Calibration(C.Structure):  # orb_calibration
    _fields_ = [(n, C.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class CamFeatures(C.Structure):  # orbm_cam_features
    _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("n", C.c_int32), ("d_depth", C.c_void_p),
                ("depth_stride", C.c_int32)]


class FImage(C.Structure):  # orbf_image
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("on_device", C.c_int32), ("generation", C.c_uint64)]


class FResult(C.Structure):  # orbf_result
    _fields_ = [("n_cams", C.c_int32), ("n_total", C.c_int32), ("counts", C.c_void_p), ("kps", C.c_void_p),
                ("desc", C.c_void_p), ("uright", C.c_void_p), ("depth", C.c_void_p), ("nmatches", C.c_int32),
                ("match_of_feature", C.c_void_p), ("cross_best_idx", C.c_void_p), ("cross_best_dist", C.c_void_p),
                ("cross_second_dist", C.c_void_p), ("gpu_wait_us", C.c_float), ("n_queries", C.c_int32),
                ("queries", C.c_void_p), ("un_x", C.c_void_p), ("un_y", C.c_void_p), ("host_us", C.c_float * 4),
                ("rig_cams", C.c_int32), ("rig_counts", C.c_void_p)]


class FMotion(C.Structure):  # orbf_motion
    _fields_ = [("du", C.c_float), ("dv", C.c_float), ("th", C.c_float)]


class FStreamStats(C.Structure):  # orbf_stream_stats
    _fields_ = [("features", C.c_int64), ("temporal_matches", C.c_int64), ("cross_accepted", C.c_int64), ("digest", C.c_uint64),
                ("seconds", C.c_double)]


class FrameDesc(C.Structure):  # orbm_frame_desc
    _fields_ = [("n_total", C.c_int32), ("n_cams", C.c_int32), ("un_x", C.c_void_p), ("un_y", C.c_void_p),
                ("octave", C.c_void_p), ("angle", C.c_void_p), ("uright", C.c_void_p), ("cam_of", C.c_void_p),
                ("local_of", C.c_void_p), ("desc", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float),
                ("max_x", C.c_float), ("max_y", C.c_float)]


class DeviceFeatures(C.Structure):  # orbf_device_features
    _fields_ = [("n_total", C.c_int32), ("n_cams", C.c_int32), ("counts", C.c_int32 * 8), ("d_desc", C.c_void_p), ("d_angle", C.c_void_p),
                ("d_un_x", C.c_void_p), ("d_un_y", C.c_void_p), ("d_octave", C.c_void_p), ("d_uright", C.c_void_p), ("stream", C.c_void_p)]


class DeviceSide(C.Structure):  # orbv_device_side
    _fields_ = [("n", C.c_int32), ("d_desc", C.c_void_p), ("d_angle", C.c_void_p), ("d_x", C.c_void_p), ("d_y", C.c_void_p),
                ("d_octave", C.c_void_p), ("d_uright", C.c_void_p), ("n_cams", C.c_int32), ("cam_start", C.c_int32 * 9)]


class BowSide(C.Structure):  # orbv_side
    _fields_ = [("n", C.c_int32), ("desc", C.c_void_p), ("angle", C.c_void_p), ("flags", C.c_void_p), ("n_nodes", C.c_int32),
                ("node_id", C.c_void_p), ("node_start", C.c_void_p), ("items", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
                ("octave", C.c_void_p), ("cam_of", C.c_void_p)]


class Triangulation(C.Structure):  # orbv_triangulation
    _fields_ = [("n_cams", C.c_int32), ("n_levels", C.c_int32), ("F12", (C.c_float * 9) * 8), ("ex", C.c_float * 8),
                ("ey", C.c_float * 8), ("scale_factors", C.c_void_p), ("level_sigma2", C.c_void_p)]


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libmorb.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbError(ORB_E_NO_DEVICE, "HIP library %s is not built (run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` or `make -C multi_orb_slam_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.orb_last_error.restype = C.c_char_p
    L.orbx_tables.argtypes = [vp] * 7
    L.orbx_create.argtypes = [vp, i32, i32, i32, i32, vp]
    L.orbx_destroy.argtypes = [vp]; L.orbx_destroy.restype = None
    L.orbx_extract.argtypes = [vp, i32] + [vp] * 8
    L.orbx_upload.argtypes = [vp, i32, vp, i32, i32, i32]
    L.orbx_upload_device.argtypes = [vp, i32, vp, i32, i32, i32]
    L.orbx_run.argtypes = [vp]
    L.orbx_count.argtypes = [vp, i32]
    L.orbx_download.argtypes = [vp, i32, vp, vp, i32]
    L.orbx_device_keypoints.argtypes = [vp, i32]; L.orbx_device_keypoints.restype = vp
    L.orbx_device_descriptors.argtypes = [vp, i32]; L.orbx_device_descriptors.restype = vp
    L.orbx_stream.argtypes = [vp]; L.orbx_stream.restype = vp
    L.orbx_wait_for_stream.argtypes = [vp, vp]
    L.orbx_bind_output.argtypes = [vp, i32, vp, vp, i32]
    L.orbx_debug_level.argtypes = [vp, i32, i32, vp, i32, vp, vp]
    L.orbx_debug_candidates.argtypes = [vp, i32, i32, vp, i32, vp]
    L.orbx_debug_distribute_octree.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]
    L.orbx_set_profiling.argtypes = [vp, i32]
    L.orbx_stage_times_us.argtypes = [vp, vp]
    L.orbm_create.argtypes = [i32, vp]
    L.orbm_destroy.argtypes = [vp]; L.orbm_destroy.restype = None
    L.orbm_stream.argtypes = [vp]; L.orbm_stream.restype = vp
    L.orbm_descriptor_distance.argtypes = [vp, vp]
    L.orbm_three_maxima.argtypes = [vp, i32, vp]; L.orbm_three_maxima.restype = None
    L.orbm_hamming_top2.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbm_use_matrix_cores.argtypes = [i32]
    L.orbm_use_fp4_top2.argtypes = [i32]
    L.orbm_top2_scratch_bytes.argtypes = [i32, i32]; L.orbm_top2_scratch_bytes.restype = C.c_size_t
    L.orbm_hamming_top2_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.orbm_hamming_matrix.argtypes = [vp, vp, i32, vp, i32, vp]
    L.orbm_hamming_matrix_device.argtypes = [vp, i32, vp, i32, vp, vp]
    L.orbm_frame_create.argtypes = [vp, vp, vp]
    L.orbm_frame_create_resident.argtypes = [vp, vp, vp, vp]
    L.orbm_frame_destroy.argtypes = [vp]; L.orbm_frame_destroy.restype = None
    L.orbm_set_stream.argtypes = [vp, vp]
    L.orbm_wait_for_stream.argtypes = [vp, vp]
    L.orbm_frame_from_device.argtypes = [vp, vp, i32, f32, f32, f32, f32, f32, vp]
    L.orbm_frame_download.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbm_frame_count.argtypes = [vp]
    L.orbx_debug_last_path.argtypes = [vp]
    L.orbx_debug_level0_in_place.argtypes = [vp]
    L.orbx_debug_pyramid_form.argtypes = [vp]
    L.orbx_debug_pyramid_plan.argtypes = [i32, i32, i32]
    L.orbf_create.argtypes = [vp, i32, i32, i32, i32, vp]
    L.orbf_create_depth.argtypes = [vp, i32, i32, i32, i32, i32, vp]
    L.orbf_destroy.argtypes = [vp]; L.orbf_destroy.restype = None
    L.orbf_set_depth.argtypes = [vp, i32, vp, i32]
    L.orbf_configure.argtypes = [vp, f32, i32, i32]
    L.orbf_step.argtypes = [vp, vp, vp, i32, i32, vp]
    L.orbf_step_motion.argtypes = [vp, vp, vp, i32, vp]
    L.orbf_reset.argtypes = [vp]
    L.orbf_run_stream.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, C.c_float, vp]
    L.orbf_prefetch.argtypes = [vp, vp]
    L.orbf_step_begin.argtypes = [vp, vp, vp, i32, i32, vp]
    L.orbf_step_motion_begin.argtypes = [vp, vp, vp, i32, vp]
    L.orbf_step_end.argtypes = [vp, vp]
    L.orbm_cross_top2_gathered_enqueue.argtypes = [vp, vp, i32, C.c_size_t, i32, i32, i32, vp, i32]
    L.orbm_cross_top2_gathered_collect.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbf_export_block.argtypes = [vp, vp, vp, vp]
    L.orbf_exchange_unique_id.argtypes = [vp]
    L.orbf_exchange_init.argtypes = [vp, vp, i32, i32]
    L.orbf_exchange_active.argtypes = [vp]
    L.orbf_step_motion_ahead.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp, vp]
    L.orbf_ahead_depth.argtypes = [vp]
    L.orbf_exchange_placement.argtypes = [vp]
    L.orbf_debug_exchange_timing.argtypes = [vp, C.c_int]
    L.orbf_debug_exchange_us.argtypes = [vp, C.POINTER(C.c_float)]
    L.orbf_debug_exchange_redos.argtypes = [vp]; L.orbf_debug_exchange_redos.restype = C.c_long
    L.orbf_exchange_init_loopback.argtypes = [vp, i32, i32, i32]
    L.orbf_exchange_shutdown.argtypes = [vp]
    L.orbf_exchange_peer_handle_bytes.argtypes = []; L.orbf_exchange_peer_handle_bytes.restype = C.c_size_t
    L.orbf_exchange_peer_export.argtypes = [vp, i32, i32, vp]
    L.orbf_exchange_peer_open.argtypes = [vp, vp]
    L.orbf_peek_block.argtypes = [vp, vp, vp, vp, vp]
    L.orbm_cross_top2_gathered_views.argtypes = [vp, vp, vp, vp]
    L.orbm_cross_top2_gathered.argtypes = [vp, vp, i32, C.c_size_t, i32, i32, i32, vp, vp, vp, vp, vp]
    L.orbf_extractor.argtypes = [vp]; L.orbf_extractor.restype = vp
    L.orbf_matcher.argtypes = [vp]; L.orbf_matcher.restype = vp
    L.orbm_debug_last_resolve.argtypes = [vp, vp]
    L.orbm_queries_from_motion.argtypes = [vp, vp, vp, vp, i32, f32, f32, f32, vp, f32, vp, vp, vp]
    L.orbm_count_ratio_accepted.argtypes = [vp, vp, i32, i32, f32]
    L.orbm_set_calibration.argtypes = [vp, vp]
    L.orbm_undistort_points.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orbm_image_bounds.argtypes = [vp, i32, i32, vp]
    L.orbf_set_calibration.argtypes = [vp, vp]
    L.orbm_cross_top2.argtypes = [vp, vp, vp, vp, vp]
    L.orbm_cross_top2_blocks.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.orbm_frame_grid.argtypes = [vp, vp, vp]
    L.orbm_features_in_area.argtypes = [vp, vp, i32, f32, f32, f32, i32, i32, vp, i32, vp]
    L.orbm_project_candidates.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.orbm_project_best.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, vp, vp]
    L.orbm_search_by_projection.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp]
    L.orbm_search_by_projection_windows.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, vp, vp]
    L.orbm_debug_time_project.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.orbm_search_by_projection_points.argtypes = [vp, vp, vp, i32, vp, f32, i32, vp, vp]
    f64 = C.c_double
    L.orbv_create.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp]
    L.orbv_load_text.argtypes = [C.c_char_p, i32, vp]
    L.orbv_destroy.argtypes = [vp]; L.orbv_destroy.restype = None
    L.orbv_info.argtypes = [vp, vp, vp, vp, vp]
    L.orbv_stream.argtypes = [vp]; L.orbv_stream.restype = vp
    L.orbv_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbv_transform_device.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.orbv_bow_vectors.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.orbv_score_l1.argtypes = [vp, vp, i32, vp, vp, i32]; L.orbv_score_l1.restype = f64
    L.orbv_workspace_create.argtypes = [i32, vp]
    L.orbv_workspace_destroy.argtypes = [vp]; L.orbv_workspace_destroy.restype = None
    L.orbv_search_by_bow.argtypes = [vp, vp, vp, i32, i32, f32, i32, vp, vp]
    L.orbv_search_for_triangulation.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.orbv_keyframe_create.argtypes = [vp, vp, vp]
    L.orbv_keyframe_destroy.argtypes = [vp]; L.orbv_keyframe_destroy.restype = None
    L.orbv_keyframe_count.argtypes = [vp]
    L.orbv_keyframe_from_device.argtypes = [vp, vp, vp, i32, vp, vp]
    L.orbv_keyframe_download.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbf_export_features.argtypes = [vp, vp]
    L.orbv_search_by_bow_resident.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, i32, vp, vp]
    L.orbv_search_for_triangulation_resident.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != ORB_OK:
        raise OrbError(rc, lib().orb_last_error().decode("utf-8", "replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
