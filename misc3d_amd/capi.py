"""ctypes binding of the C ABI declared in ``include/misc3d_amd.h``.

This is plumbing for the tests, ``bench.py`` and the distributed driver: it loads
``misc3d_amd/lib/libmisc3d_amd.so`` (built in-tree by ``misc3d_amd/csrc/Makefile``) and exposes each
``m3d_*`` entry point with numpy in/out.  It contains no arithmetic and no fallback: if the shared
object is missing, importing fails loudly; if there is no HIP device every compute call raises
``M3DError`` carrying the library's message.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3D_FP_ORDER=1|2 in the environment loads the library built for one of the alternative floating-point associations
# (misc3d_amd/csrc/m3d_fp.hpp; tests/test_fp_orders.py runs the parity suite under each); default 0
FP_ORDER = int(os.environ.get("M3D_FP_ORDER", "0") or 0)
# M3D_LIB_VARIANT=<dir>: a diagnostic build under misc3d_amd/lib/<dir>/ (tools/reg_query_fates.sh: -DM3D_REG_TRIP_STATS)
_VARIANT = os.environ.get("M3D_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "lib", *([_VARIANT] if _VARIANT else ([f"order{FP_ORDER}"] if FP_ORDER else [])), "libmisc3d_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "misc3d_amd.h")

PLANE, SPHERE, CYLINDER = 0, 1, 2
MINIMAL_SAMPLE = {PLANE: 3, SPHERE: 4, CYLINDER: 2}
NUM_PARAMS = {PLANE: 4, SPHERE: 4, CYLINDER: 7}
MODEL_STRIDE = 8

OK, FALSE = 1, 0
ERR_PROBABILITY, ERR_TOO_FEW_POINTS, ERR_NO_NORMALS, ERR_SIZE_MISMATCH = -1, -2, -3, -4
ERR_INVALID_ARG, ERR_DEVICE, ERR_INTERNAL = -5, -6, -7


class M3DError(RuntimeError):
    """Raised for negative return codes; mirrors misc3d::LogError -> std::runtime_error
    (src/logging.cpp:64-74), message prefixed like the reference's."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[Misc3D Error] {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("fitness", C.c_double), ("inlier_rmse", C.c_double), ("count", C.c_uint64),
                ("iterations", C.c_uint64), ("best_index", C.c_int64), ("general_fit_ok", C.c_int32),
                ("ties", C.c_int32), ("hypotheses_scored", C.c_uint64), ("exact_rmse_evals", C.c_uint64),
                ("ms_sample", C.c_double), ("ms_score", C.c_double), ("ms_refine", C.c_double),
                ("ms_total", C.c_double), ("ms_score_kernel", C.c_double), ("score_launches", C.c_uint32),
                ("early_pick_redone", C.c_uint32), ("pairs_scored", C.c_uint64), ("pairs_exact", C.c_uint64), ("pairs_timed", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RegStats(C.Structure):
    _fields_ = [("fitness", C.c_double), ("inlier_rmse", C.c_double), ("validations", C.c_uint64),
                ("iterations", C.c_int64), ("best_index", C.c_int64), ("est_k", C.c_int64),
                ("ms_total", C.c_double), ("ties", C.c_uint64), ("exact_rmse_evals", C.c_uint64),
                ("lds_wave_hypotheses", C.c_uint64), ("global_wave_hypotheses", C.c_uint64),
                ("nn_fp32_screen", C.c_uint64), ("nn_screen_fallbacks", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ReplayState(C.Structure):
    _fields_ = [("best_fitness", C.c_double), ("best_rmse", C.c_double), ("best_index", C.c_int64),
                ("best_count", C.c_uint64), ("count", C.c_uint64), ("current_iteration", C.c_uint64),
                ("iterations", C.c_uint64), ("best_rmse_known", C.c_int32), ("stopped", C.c_int32)]


RMSE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_size_t)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C misc3d_amd/csrc` (or __graft_entry__.build()); "
                "misc3d_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.m3d_last_error.restype = C.c_char_p
        L.m3d_version.restype = C.c_char_p
        L.m3d_cloud_create.restype = C.c_void_p
        L.m3d_cloud_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.m3d_cloud_destroy.argtypes = [C.c_void_p]
        L.m3d_cloud_destroy.restype = None
        L.m3d_host_alloc.restype = C.c_void_p
        L.m3d_host_alloc.argtypes = [C.c_size_t]
        L.m3d_host_free.restype = None
        L.m3d_host_free.argtypes = [C.c_void_p]
        L.m3d_cloud_size.argtypes = [C.c_void_p]
        L.m3d_minimal_fit.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_cloud_size.restype = C.c_size_t
        L.m3d_cloud_fit.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_double, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_cloud_score_range.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_size_t,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_cloud_exact_error.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_cloud_refine.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_cloud_refine_expect.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_void_p]
        L.m3d_cloud_remove_inliers.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.m3d_cloud_original_size.restype = C.c_size_t
        L.m3d_cloud_original_size.argtypes = [C.c_void_p]
        L.m3d_bench_time_score.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p]     # include/misc3d_amd_bench.h
        L.m3d_bench_plane_upper_bounds.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p]
        L.m3d_bench_upper_bounds.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p]
        L.m3d_sampler_create.restype = C.c_void_p
        L.m3d_sampler_create.argtypes = [C.c_size_t, C.c_int, C.c_uint64]
        L.m3d_sampler_destroy.argtypes = [C.c_void_p]
        L.m3d_sampler_destroy.restype = None
        L.m3d_sampler_drawn.argtypes = [C.c_void_p]
        L.m3d_sampler_drawn.restype = C.c_size_t
        L.m3d_sampler_table.argtypes = [C.c_void_p, C.c_size_t]
        L.m3d_sampler_table.restype = C.c_void_p
        L.m3d_cloud_score_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_draw_samples.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_uint64, C.c_void_p]
        L.m3d_replay_init.argtypes = [C.c_void_p]
        L.m3d_replay_init.restype = None
        L.m3d_replay_chunk.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_double, C.c_size_t,
                                       C.c_size_t, C.c_void_p, C.c_void_p, RMSE_FN, C.c_void_p]
        L.m3d_replay_chunk.restype = None
        for name in ("m3d_fit_plane", "m3d_fit_sphere"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_size_t, C.c_double, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_fit_cylinder.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_size_t, C.c_double,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_segment_plane_iterative.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double,
                                                  C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]
        L.m3d_kabsch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.m3d_registration_ransac.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double, C.c_double,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.m3d_reg_create.restype = C.c_void_p
        L.m3d_reg_create.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int]
        L.m3d_reg_destroy.restype = None
        L.m3d_reg_destroy.argtypes = [C.c_void_p]
        L.m3d_reg_begin_chunk.argtypes = [C.c_void_p, C.c_void_p]
        L.m3d_reg_validate.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.m3d_reg_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_reg_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_registration_icp.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_int,
                                           C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_normals_from_map.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p]
        L.m3d_information_matrix.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p]
        L.m3d_detect_boundary_points.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_double,
                                                 C.c_int, C.c_void_p, C.c_void_p]
        L.m3d_match_last_fallbacks.restype = C.c_uint64
        L.m3d_match_last_fallbacks.argtypes = []
        L.m3d_match_mutual_nn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        # multi-GPU: communicators and the sharded entry points
        L.m3d_comm_unique_id.argtypes = [C.c_void_p]
        L.m3d_comm_create_rccl.restype = C.c_void_p
        L.m3d_comm_create_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.m3d_comm_create_host.restype = C.c_void_p
        L.m3d_comm_create_host.argtypes = [C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]
        L.m3d_comm_create_local.argtypes = [C.c_int, C.c_void_p]
        L.m3d_comm_destroy.restype = None
        L.m3d_comm_destroy.argtypes = [C.c_void_p]
        L.m3d_comm_world.argtypes = [C.c_void_p]
        L.m3d_comm_rank.argtypes = [C.c_void_p]
        L.m3d_comm_collectives.restype = C.c_uint64
        L.m3d_comm_collectives.argtypes = [C.c_void_p]
        L.m3d_cloud_fit_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_double, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_segment_plane_iterative_clouds.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double,
                                                          C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_segment_plane_iterative_sharded.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double,
                                                          C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_segment_plane_iterative_multi.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double,
                                                        C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_fit_multi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_size_t, C.c_double,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.m3d_registration_ransac_sharded.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                      C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double,
                                                      C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                      C.c_void_p]
        L.m3d_bench_fp64_issue_rate.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.m3d_bench_cloud_setup_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.m3d_bench_last_segment_ms.argtypes = [C.c_void_p]
        L.m3d_bench_reg_checkers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        L.m3d_get_config.restype = None
        L.m3d_get_config.argtypes = [C.c_void_p]
        L.m3d_set_config.argtypes = [C.c_void_p]
        L.m3d_release_cached.restype = None
        L.m3d_release_cached.argtypes = [C.c_int]
        _lib = L
    return _lib


class Config(C.Structure):
    """m3d_config (include/misc3d_amd.h): the library's tunables."""
    _fields_ = [(k, C.c_int32) for k in ("dense_scoring", "speculative_refine", "lead_hypotheses",
                                         "score_groups_per_block", "score_min_workgroups", "dense_workgroups",
                                         "reg_neighbour_lists", "reg_prune", "match_brute", "match_fp32_screen",
                                         "pool_limit_mb", "kernel_timing", "reg_sorted_lists", "score_fp32_screen",
                                         "cull_fp32", "reg_fp32_screen", "sorted_tombstones", "score_mfma", "score_mfma_groups", "score_waves4", "score_waves4_groups", "score_phases", "compact_one_pass", "plane_bound",
                                         "lanes", "wait_spin_us", "prestream", "chunk_cap", "first_chunk", "reg_cells_per_radius", "match_pipeline", "reg_cache", "device_aliases", "lanes_eager")]


def fp64_issue_rate(device=0, ms_target=2.0):
    """m3d_bench_fp64_issue_rate (include/misc3d_amd_bench.h) -> (1e12 fp64 lane-ops/s the device sustains, ms measured)"""
    t, ms = C.c_double(0), C.c_double(0)
    _check(lib().m3d_bench_fp64_issue_rate(device, ms_target, C.cast(C.byref(t), C.c_void_p), C.cast(C.byref(ms), C.c_void_p)))
    return float(t.value), float(ms.value)


def mfma_probe(xyz512, box, max_abs, records, device=0):
    """m3d_bench_mfma_probe (include/misc3d_amd_bench.h): the MFMA screen's values for one tile of 512 points and plane
    records (a, b, c, d, T) -> (u[n_h, 512, 2] = the pipe's (T - S, T + S), h[n_h] = the band on their product, sigma[n_h],
    e_p[n_h] = the bound on either u, offsets32[512, 3])"""
    xyz = np.ascontiguousarray(xyz512, dtype=np.float64).reshape(512, 3)
    bx = np.ascontiguousarray(box, dtype=np.float64).reshape(6)
    rec = np.zeros((len(records), 8), dtype=np.float64)
    rec[:, :5] = np.asarray(records, dtype=np.float64).reshape(-1, 5)
    q = np.empty((len(rec), 512, 2), dtype=np.float64)
    h = np.empty((len(rec), 3), dtype=np.float64)
    off = np.empty((512, 3), dtype=np.float32)
    f = lib().m3d_bench_mfma_probe
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(f(device, xyz.ctypes.data, bx.ctypes.data, float(max_abs), rec.ctypes.data, len(rec), q.ctypes.data, h.ctypes.data,
             off.ctypes.data))
    return q, h[:, 0].copy(), h[:, 1].copy(), h[:, 2].copy(), off


def match_last_path() -> int:
    """m3d_bench_match_last_path: bit 0 MFMA screen, 1 sliced uploads, 2 redone whole, 3 fp32 screen, 4 brute force (the calling thread's last match)"""
    f = lib().m3d_bench_match_last_path
    f.restype = C.c_uint
    f.argtypes = []
    return int(f())


def experimental() -> bool:
    """m3d_bench_experimental: was the library built with -DM3D_EXPERIMENTAL (round 4's refuted variants compiled in)?"""
    f = lib().m3d_bench_experimental
    f.restype = C.c_int
    f.argtypes = []
    return bool(f())


def get_config() -> Config:
    c = Config()
    lib().m3d_get_config(C.byref(c))
    return c


def set_config(**kw) -> Config:
    """Change some tunables (m3d_set_config); returns the PREVIOUS settings, to be restored with restore_config."""
    old = get_config()
    new = get_config()
    for k, v in kw.items():
        if not hasattr(new, k):
            raise AttributeError(k)
        setattr(new, k, int(v))
    _check(lib().m3d_set_config(C.byref(new)))
    return old


def restore_config(cfg: Config):
    _check(lib().m3d_set_config(C.byref(cfg)))


class Comm:
    """m3d_comm: the exchange of the hypothesis-sharded entry points (include/misc3d_amd.h).

    Comm.rccl(group, device)   one process per GPU: rank 0's ncclUniqueId travels over `group` (any
                               torch.distributed group, e.g. gloo), then ncclCommInitRank inside the library.
    Comm.host(world, rank, fn) fn(send: bytes) -> bytes of world * len(send): caller-supplied all-gather.
    Comm.torch_host(group)     the same over torch.distributed.all_gather_into_tensor of `group` (tests: gloo)."""

    def __init__(self, handle, keep=None):
        if not handle:
            raise M3DError(ERR_DEVICE, last_error())
        self._h = handle
        self._keep = keep

    @classmethod
    def rccl(cls, group=None, device=0, world=None, rank=None):
        import torch
        import torch.distributed as dist
        if world is None:
            world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_initialized() else (1, 0)
        ident = np.zeros(128, dtype=np.uint8)
        status = np.zeros(129, dtype=np.uint8)     # the id + one byte that says rank 0 got one (nobody waits for a rank that raised)
        err = None
        if rank == 0:
            try:
                _check(lib().m3d_comm_unique_id(_p(ident)))
                status[:128] = ident
                status[128] = 1
            except M3DError as e:
                err = e
        if world > 1:
            t = torch.from_numpy(status)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not status[128]:
            raise err if err is not None else M3DError(-6, "rank 0 could not create an RCCL unique id")
        ident = np.ascontiguousarray(status[:128])
        # librccl prints a version banner to STDOUT on its first communicator; callers (bench.py) own stdout, so the
        # banner is sent to stderr: fd 1 points at fd 2 for the duration of the call, C stdio flushed on both sides.
        # SIDE EFFECT: the redirection is process-wide and not thread-safe -- whatever another thread writes
        # to stdout while ncclCommInitRank runs (it can take seconds) lands on stderr; M3D_RCCL_KEEP_STDOUT=1 skips it
        # (the banner then goes where librccl sends it)
        libc = C.CDLL(None)
        sys.stdout.flush()
        libc.fflush(None)
        if os.environ.get("M3D_RCCL_KEEP_STDOUT") == "1":
            return cls(lib().m3d_comm_create_rccl(_p(ident), world, rank, device))
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            h = lib().m3d_comm_create_rccl(_p(ident), world, rank, device)
        finally:
            libc.fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        return cls(h)

    @classmethod
    def host(cls, world, rank, fn):
        def _cb(_user, send, recv, nbytes):
            try:
                out = fn(C.string_at(send, nbytes))
                if len(out) != nbytes * world:
                    return 2
                C.memmove(recv, out, nbytes * world)
                return 0
            except Exception:       # noqa: BLE001 -- must not unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLGATHER_FN(_cb)
        return cls(lib().m3d_comm_create_host(world, rank, cb, None), keep=cb)

    @classmethod
    def torch_host(cls, group=None):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def gather(b):
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
            out = torch.empty(world * len(b), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, t, group=group)
            return out.numpy().tobytes()
        return cls.host(world, rank, gather)

    @property
    def world(self):
        return int(lib().m3d_comm_world(self._h))

    @property
    def rank(self):
        return int(lib().m3d_comm_rank(self._h))

    @property
    def collectives(self):
        return int(lib().m3d_comm_collectives(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().m3d_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def last_error() -> str:
    return lib().m3d_last_error().decode()


def device_count() -> int:
    return int(lib().m3d_device_count())


def _check(rc: int) -> int:
    if rc < 0:
        raise M3DError(rc, last_error())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _addr(a):
    """address of a (contiguous) array for a c_void_p parameter, or None"""
    return a.__array_interface__["data"][0] if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _seed_ref(seed):
    if seed is None:
        return None, None
    s = C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)
    return s, C.byref(s)


class Fit:
    """ret: 1 = reference `true`, 0 = reference `false`; params: best model AFTER RefineModel (not zeroed; callers zero on
    ret == 0); inliers: uint64, ascending; stats: m3d_stats as a dict (+ "n_inliers"), built when first asked for."""
    __slots__ = ("ret", "params", "inliers", "_stats", "_raw")

    def __init__(self, ret, params, inliers, stats):
        self.ret = ret
        self.params = params
        self.inliers = inliers
        if isinstance(stats, dict):
            self._stats, self._raw = stats, None
        else:   # (Stats structure, inlier count)
            self._stats, self._raw = None, stats

    @property
    def stats(self) -> dict:
        if self._stats is None:
            st, ni = self._raw
            self._stats = st.asdict()
            self._stats["n_inliers"] = int(ni)
            self._raw = None
        return self._stats

    def __iter__(self):   # (ret, params, inliers, stats), as the dataclass this used to be unpacked
        return iter((self.ret, self.params, self.inliers, self.stats))


class Sampler:
    """m3d_sampler: the sequential RandomSampler (utils.h:71-97) with an explicit seed; records its table."""

    def __init__(self, n_points, kind, seed):
        self.kind = kind
        self.m = MINIMAL_SAMPLE[kind]
        self._h = lib().m3d_sampler_create(n_points, kind, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF))
        if not self._h:
            raise M3DError(ERR_INVALID_ARG, last_error())

    def table(self, n_hyp):
        """(n_hyp, m) uint32 view of the first n_hyp samples (drawing them if necessary)."""
        ptr = lib().m3d_sampler_table(self._h, n_hyp)
        if n_hyp == 0:
            return np.zeros((0, self.m), dtype=np.uint32)
        buf = (C.c_uint32 * (n_hyp * self.m)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(n_hyp, self.m)

    @property
    def drawn(self):
        return int(lib().m3d_sampler_drawn(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().m3d_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_layout(begin, end, slice_, world):
    """Index bookkeeping of m3d_cloud_score_shard: for the window [begin, end) cut into slices of
    `slice_`, returns (mine, order) where mine[r] = global indices owned by rank r (ascending) and the
    global order is recovered from the rank-major concatenation (padded to max_mine per rank) through
    `order`: global_records = gathered.reshape(world, max_mine)[order_rank, order_pos]."""
    g = np.arange(begin, end, dtype=np.int64)
    j = (g - begin) // slice_
    owner = (j % world).astype(np.int64)
    pos = np.empty(len(g), dtype=np.int64)
    counts = np.zeros(world, dtype=np.int64)
    for r in range(world):
        sel = owner == r
        k = int(sel.sum())
        pos[sel] = np.arange(k)
        counts[r] = k
    return owner, pos, counts


class _PinnedU64:
    """n uint64 of page-locked host memory from m3d_host_alloc, exposed through the array interface so that numpy
    arrays made from it (and their views) keep it alive."""

    def __init__(self, n):
        self._ptr = lib().m3d_host_alloc(8 * n)
        if not self._ptr:
            raise MemoryError(last_error())
        self.__array_interface__ = {"shape": (n,), "typestr": "<u8", "data": (self._ptr, False), "version": 3}

    def __del__(self):
        try:
            if self._ptr:
                lib().m3d_host_free(self._ptr)
                self._ptr = None
        except Exception:
            pass


class Cloud:
    """Resident SoA copy of a point cloud in HBM (RANSAC::SetPointCloud, ransac.h:469-475)."""

    def make_sampler(self, kind, seed):
        return Sampler(self.n, kind, seed)

    def score_shard_packed(self, sampler, threshold, begin, end, slice_, world, rank):
        """m3d_cloud_score_shard with valid == NULL -> uint32 records (valid << 31 | count) of this rank's hypotheses."""
        rec = np.empty(max(end - begin, 1), dtype=np.uint32)
        n = C.c_size_t(0)
        _check(lib().m3d_cloud_score_shard(self._h, sampler._h, threshold, begin, end, slice_, world, rank, _p(rec),
                                           None, C.cast(C.byref(n), C.c_void_p)))
        return rec[: n.value]

    def score_shard(self, sampler, threshold, begin, end, slice_, world, rank):
        """m3d_cloud_score_shard -> (valid, counts) of this rank's hypotheses in [begin, end)."""
        cap = end - begin
        cnt = np.empty(max(cap, 1), dtype=np.uint32)
        val = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        _check(lib().m3d_cloud_score_shard(self._h, sampler._h, threshold, begin, end, slice_, world, rank, _p(cnt),
                                           _p(val), C.cast(C.byref(n), C.c_void_p)))
        return val[: n.value], cnt[: n.value]

    def __init__(self, xyz, normals=None, device: int = 0, lane: int = -1):
        """lane >= 0: m3d_cloud_create_lane -- the cloud lives on THAT lane of the device (a single-threaded caller spreads its
        clouds over the lanes so that fit_batch can run their fits side by side); -1: the calling thread's own lane."""
        xyz = _f64(xyz).reshape(-1, 3)
        nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
        if nrm is not None and len(nrm) != len(xyz):
            raise ValueError("normals and points differ in length")
        self.n = len(xyz)            # points currently in the cloud (shrinks with remove_inliers)
        self.n_created = len(xyz)    # index lists refer to the cloud as created
        self._host_xyz, self._host_nrm = xyz, nrm     # for minimal_model (host-side MinimalFit of one sample)
        if lane >= 0:
            f = lib().m3d_cloud_create_lane
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
            self._h = f(_p(xyz), _p(nrm), self.n, device, int(lane))
        else:
            self._h = lib().m3d_cloud_create(_p(xyz), _p(nrm), self.n, device)
        if not self._h:
            raise M3DError(ERR_DEVICE, last_error())

    def setup_ms(self):
        """m3d_bench_cloud_setup_ms -> dict(total, upload_transpose, bbox, sort, tile_boxes) in ms (the phases are filled
        when m3d_config.kernel_timing was set at creation)."""
        out = np.zeros(5)
        _check(lib().m3d_bench_cloud_setup_ms(self._h, _p(out)))
        return dict(zip(("total", "upload_transpose", "bbox", "sort", "tile_boxes"), (float(v) for v in out)))

    def _out_buf(self):
        """Per-cloud index-list buffer (n_created uint64), page-locked through m3d_host_alloc when possible: the
        library then copies the inlier list while the GeneralFit sums run (include/misc3d_amd.h).  The block is
        owned by the array (and every view of it, e.g. fit(copy=False).inliers): it is released when the last
        one goes away, not by close()."""
        if getattr(self, "_inl_buf", None) is None:
            n = max(self.n_created, 1)
            try:
                self._inl_buf = np.asarray(_PinnedU64(n))
            except MemoryError:
                self._inl_buf = np.empty(n, dtype=np.uint64)
        return self._inl_buf

    def close(self):
        if getattr(self, "_h", None):
            lib().m3d_cloud_destroy(self._h)
            self._h = None
        self._inl_buf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def fit(self, kind, threshold=0.01, max_iteration=1000, probability=0.9999, seed=None,
            want_inliers=True, copy=True) -> Fit:
        """m3d_cloud_fit.  copy=False returns a view of a per-Cloud output buffer that the next
        call overwrites (the C ABI writes into caller-allocated memory; this avoids re-allocating
        8 bytes x N per call)."""
        params = np.zeros(NUM_PARAMS[kind])
        inl = None
        if want_inliers:
            # copy=True: the list is copied out anyway -- the thread's page-locked scratch serves every cloud (a buffer per
            # cloud costs a hipHostMalloc of 8 bytes x N: 14 ms for 10 M points, paid by the first fit on it)
            inl = _seg_scratch(max(self.n_created, 1)) if copy else None
            if inl is None:
                inl = self._out_buf()
        ni = C.c_size_t(0)
        st = Stats()
        _s, sref = _seed_ref(seed)
        # (byref objects and plain addresses go straight into c_void_p parameters: ndarray.ctypes.data_as and ctypes.cast
        #  cost 2.4 and 0.7 us apiece, a third of this wrapper's time around a 0.27 ms fit)
        rc = _check(lib().m3d_cloud_fit(self._h, kind, threshold, max_iteration, probability, sref, _addr(params),
                                        _addr(inl), C.byref(ni), C.byref(st)))
        if want_inliers:
            inliers = inl[: ni.value].copy() if copy else inl[: ni.value]
        else:
            inliers = np.zeros(0, dtype=np.uint64)
        return Fit(rc, params, inliers, (st, ni.value))

    def fit_sharded(self, comm, kind, threshold=0.01, max_iteration=1000, probability=0.9999, seed=None,
                    want_inliers=True, copy=True) -> Fit:
        """m3d_cloud_fit_sharded: the same fit with the hypothesis loop sharded over the ranks of `comm` (every
        rank calls it on its own replica of the cloud and gets the same result)."""
        params = np.zeros(NUM_PARAMS[kind])
        inl = self._out_buf() if want_inliers else None
        ni = C.c_size_t(0)
        st = Stats()
        _s, sref = _seed_ref(seed)
        rc = _check(lib().m3d_cloud_fit_sharded(self._h, comm._h if comm is not None else None, kind, threshold,
                                                max_iteration, probability, sref, _addr(params), _addr(inl),
                                                C.byref(ni), C.byref(st)))
        if want_inliers:
            inliers = inl[: ni.value].copy() if copy else inl[: ni.value]
        else:
            inliers = np.zeros(0, dtype=np.uint64)
        return Fit(rc, params, inliers, (st, ni.value))

    def score_range(self, kind, threshold, samples, begin=0, end=None, want_models=True):
        """m3d_cloud_score_range -> (valid, models or None, counts) for hypotheses [begin, end)."""
        samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, MINIMAL_SAMPLE[kind])
        end = len(samples) if end is None else end
        cnt = np.empty(end - begin, dtype=np.uint32)
        val = np.empty(end - begin, dtype=np.uint8)
        mod = np.empty((end - begin, MODEL_STRIDE)) if want_models else None
        _check(lib().m3d_cloud_score_range(self._h, kind, threshold, _p(samples), begin, end, _p(cnt), _p(val),
                                           _p(mod)))
        return val, (mod[:, : NUM_PARAMS[kind]].copy() if want_models else None), cnt

    def time_score(self, kind, threshold, samples, reps=5, mode=0):
        """(average ms per launch over `reps` launches measured with HIP events, surviving
        (tile, hypothesis) pairs).  mode 0 = score_list_k, 1 = cull_k, 2 = dense score_k."""
        samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, MINIMAL_SAMPLE[kind])
        ms = C.c_double(0)
        listed = C.c_uint64(0)
        _check(lib().m3d_bench_time_score(self._h, kind, threshold, _p(samples), len(samples), reps, mode,
                                          C.cast(C.byref(ms), C.c_void_p), C.cast(C.byref(listed), C.c_void_p)))
        return float(ms.value), int(listed.value)

    def plane_upper_bounds(self, threshold, samples):
        """m3d_bench_plane_upper_bounds -> uint32 upper bound of every plane hypothesis' inlier count (plane_bound_k, nothing pruned)"""
        samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, 3)
        ub = np.zeros(len(samples), dtype=np.uint32)
        _check(lib().m3d_bench_plane_upper_bounds(self._h, threshold, _p(samples), len(samples), _p(ub)))
        return ub

    def upper_bounds(self, kind, threshold, samples):
        """m3d_bench_upper_bounds -> uint32 upper bound of every hypothesis' inlier count (plane_bound_k<kind>, nothing pruned)"""
        samples = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, MINIMAL_SAMPLE[kind])
        ub = np.zeros(len(samples), dtype=np.uint32)
        _check(lib().m3d_bench_upper_bounds(self._h, kind, threshold, _p(samples), len(samples), _p(ub)))
        return ub

    def exact_error(self, kind, threshold, model):
        model = _f64(model)
        cnt = C.c_uint64(0)
        err = C.c_double(0)
        _check(lib().m3d_cloud_exact_error(self._h, kind, threshold, _p(model), C.cast(C.byref(cnt), C.c_void_p),
                                           C.cast(C.byref(err), C.c_void_p)))
        return int(cnt.value), float(err.value)

    refine_takes_expected = True

    def refine(self, kind, threshold, params, copy=True, expected=None):
        """m3d_cloud_refine; expected = inlier count already known from the scoring records (m3d_cloud_refine_expect)"""
        params = _f64(params).copy()
        inl = self._out_buf()
        ni = C.c_size_t(0)
        rc = _check(lib().m3d_cloud_refine_expect(self._h, kind, threshold, _p(params),
                                                  -1 if expected is None else int(expected), _p(inl),
                                                  C.cast(C.byref(ni), C.c_void_p)))
        return rc, params, (inl[: ni.value].copy() if copy else inl[: ni.value])

    def minimal_model(self, kind, threshold, sample_row):
        """MinimalFit of ONE sample -> 8-double model record.  Host-side (m3d_minimal_fit, bit-identical to the
        device) while the cloud is as created; after remove_inliers the sample indexes the shrunk device
        cloud, so the one-row device launch is used."""
        row = np.ascontiguousarray(sample_row, dtype=np.uint32).reshape(-1)
        if self.n == self.n_created:
            pts = np.ascontiguousarray(self._host_xyz[row])
            nrm = np.ascontiguousarray(self._host_nrm[row]) if self._host_nrm is not None else None
            model = np.zeros(8)
            ok = C.c_uint8(0)
            _check(lib().m3d_minimal_fit(kind, _p(pts), _p(nrm), _p(model), C.cast(C.byref(ok), C.c_void_p)))
            return model[: NUM_PARAMS[kind]].copy()
        _, mod, _ = self.score_range(kind, threshold, row.reshape(1, -1), 0, 1)
        return mod[0]

    def remove_inliers(self, kind, threshold, model):
        """m3d_cloud_remove_inliers: SelectByIndex(inliers of `model`, invert=True) in place -> #removed."""
        model = _f64(model).copy()
        nr = C.c_size_t(0)
        _check(lib().m3d_cloud_remove_inliers(self._h, kind, threshold, _p(model), C.cast(C.byref(nr), C.c_void_p)))
        self.n = int(lib().m3d_cloud_size(self._h))
        return nr.value


def draw_samples(n_points, kind, n_hyp, seed):
    out = np.zeros((n_hyp, MINIMAL_SAMPLE[kind]), dtype=np.uint32)
    _check(lib().m3d_draw_samples(n_points, kind, n_hyp, C.c_uint64(seed), _p(out)))
    return out


def fit(kind, xyz, normals=None, threshold=0.01, max_iteration=1000, probability=0.9999, seed=None,
        device=0, copy=True) -> Fit:
    """One-shot m3d_fit_plane / m3d_fit_sphere / m3d_fit_cylinder.  copy=False: the inlier list is a view of the
    thread's page-locked scratch (overwritten by the next call)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    params = np.zeros(NUM_PARAMS[kind])
    # filled by the library up to n_inliers, copied out below: the thread's page-locked scratch when there is one (the
    # compaction kernel then stores the index list straight into it)
    inl = _seg_scratch(max(n, 1))
    if inl is None:
        inl = np.empty(max(n, 1), dtype=np.uint64)
    ni = C.c_size_t(0)
    st = Stats()
    _s, sref = _seed_ref(seed)
    sp = C.cast(sref, C.c_void_p) if sref else None
    nip, stp = C.cast(C.byref(ni), C.c_void_p), C.cast(C.byref(st), C.c_void_p)
    if kind == CYLINDER:
        nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
        rc = lib().m3d_fit_cylinder(_p(xyz), _p(nrm), n, threshold, max_iteration, probability, sp, device,
                                    _p(params), _p(inl), nip, stp)
    else:
        fn = lib().m3d_fit_plane if kind == PLANE else lib().m3d_fit_sphere
        rc = fn(_p(xyz), n, threshold, max_iteration, probability, sp, device, _p(params), _p(inl), nip, stp)
    _check(rc)
    d = st.asdict()
    d["n_inliers"] = int(ni.value)
    return Fit(rc, params, inl[: ni.value].copy() if copy else inl[: ni.value], d)


_seg_tls = threading.local()
_SEG_SCRATCH_MAX = 1 << 25      # uint64 entries (256 MB): larger clouds take the pageable path


def _seg_scratch(n):
    """n uint64 of page-locked scratch, kept per thread and grown on demand (None: too large, or pinning failed)."""
    if n > _SEG_SCRATCH_MAX:
        return None
    buf = getattr(_seg_tls, "buf", None)
    if buf is None or len(buf) < n:
        try:
            buf = np.asarray(_PinnedU64(n))
        except MemoryError:
            return None
        _seg_tls.buf = buf
    return buf[:n]


def segment_plane_iterative(xyz, threshold, max_iteration=100, min_ratio=0.05, seed=None, device=0,
                            max_clusters=4096, copy=True, with_points=False):
    """m3d_segment_plane_iterative (with_points: m3d_segment_plane_iterative_clouds -- a fourth return value, the list of
    the clusters' (count, 3) point arrays, views of one array gathered on the device).  copy=True (default): the library writes the index lists into a page-locked scratch
    the binding keeps per thread, every cluster is returned as an array of its own.  copy=False returns the clusters as
    views of ONE pageable index array instead (no copies, but the library then reaches the array through staged copies
    and fresh pages: on 10 M points 43 ms against 37)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    planes = np.zeros((max_clusters, 4))
    offs = np.zeros(max_clusters + 1, dtype=np.uint64)
    # filled by the library up to the last offset.  copy=True: the clusters are copied out anyway, so the library may as
    # well write into a page-locked scratch kept per thread (its kernels then store the index lists straight into it: no
    # staged copies into fresh pageable pages, which on 10 M points cost 3 ms of page faults and a blocking copy per round)
    idx = _seg_scratch(max(n, 1)) if copy else None
    if idx is None:
        idx = np.empty(max(n, 1), dtype=np.uint64)
    k = C.c_size_t(0)
    _s, sref = _seed_ref(seed)
    if with_points:
        cpts = np.empty((max(n, 1), 3))
        rc = _check(lib().m3d_segment_plane_iterative_clouds(_p(xyz), n, threshold, max_iteration, min_ratio,
                                                             C.cast(sref, C.c_void_p) if sref else None, device,
                                                             max_clusters, _p(planes), _p(offs), _p(idx), _p(cpts),
                                                             C.cast(C.byref(k), C.c_void_p)))
    else:
        rc = _check(lib().m3d_segment_plane_iterative(_p(xyz), n, threshold, max_iteration, min_ratio,
                                                      C.cast(sref, C.c_void_p) if sref else None, device, max_clusters,
                                                      _p(planes), _p(offs), _p(idx), C.cast(C.byref(k), C.c_void_p)))
    k = k.value
    clusters = [idx[int(offs[i]): int(offs[i + 1])].copy() if copy else idx[int(offs[i]): int(offs[i + 1])] for i in range(k)]
    if with_points:
        return rc, planes[:k].copy(), clusters, [cpts[int(offs[i]): int(offs[i + 1])] for i in range(k)]
    return rc, planes[:k].copy(), clusters


def last_segment_ms():
    """m3d_bench_last_segment_ms -> dict of the calling thread's last segmentation call (ms)"""
    out = np.zeros(6)
    _check(lib().m3d_bench_last_segment_ms(_p(out)))
    return {"total": out[0], "cloud_create": out[1], "rounds": out[2], "final_copy": out[3], "big_rounds": out[4],
            "n_big_rounds": int(out[5]), "n_rounds": int(round((out[5] - int(out[5])) * 1e4))}


def segment_plane_iterative_sharded(xyz, comm, threshold, max_iteration=100, min_ratio=0.05, seed=None, device=0,
                                    max_clusters=4096):
    """m3d_segment_plane_iterative_sharded (every rank of `comm` calls it with the same cloud)."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    planes = np.zeros((max_clusters, 4))
    offs = np.zeros(max_clusters + 1, dtype=np.uint64)
    idx = np.empty(max(n, 1), dtype=np.uint64)
    k = C.c_size_t(0)
    _s, sref = _seed_ref(seed)
    rc = _check(lib().m3d_segment_plane_iterative_sharded(_p(xyz), n, threshold, max_iteration, min_ratio,
                                                          C.cast(sref, C.c_void_p) if sref else None, device,
                                                          comm._h if comm is not None else None, max_clusters,
                                                          _p(planes), _p(offs), _p(idx), C.cast(C.byref(k), C.c_void_p)))
    k = k.value
    return rc, planes[:k].copy(), [idx[int(offs[i]): int(offs[i + 1])].copy() for i in range(k)]


def segment_plane_iterative_multi(xyz, devices, threshold, max_iteration=100, min_ratio=0.05, seed=None,
                                  max_clusters=4096):
    """m3d_segment_plane_iterative_multi: one process, a thread and a replica per device of `devices`."""
    xyz = _f64(xyz).reshape(-1, 3)
    n = len(xyz)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    planes = np.zeros((max_clusters, 4))
    offs = np.zeros(max_clusters + 1, dtype=np.uint64)
    idx = np.empty(max(n, 1), dtype=np.uint64)
    k = C.c_size_t(0)
    _s, sref = _seed_ref(seed)
    rc = _check(lib().m3d_segment_plane_iterative_multi(_p(xyz), n, threshold, max_iteration, min_ratio,
                                                        C.cast(sref, C.c_void_p) if sref else None, _p(dev), len(dev),
                                                        max_clusters, _p(planes), _p(offs), _p(idx),
                                                        C.cast(C.byref(k), C.c_void_p)))
    k = k.value
    return rc, planes[:k].copy(), [idx[int(offs[i]): int(offs[i + 1])].copy() for i in range(k)]


class FitJob(C.Structure):
    """m3d_fit_job (include/misc3d_amd.h)"""
    _fields_ = [("cloud", C.c_void_p), ("kind", C.c_int32), ("has_seed", C.c_int32), ("threshold", C.c_double),
                ("probability", C.c_double), ("max_iteration", C.c_uint64), ("seed", C.c_uint64), ("inliers", C.c_void_p),
                ("params", C.c_double * 8), ("n_inliers", C.c_uint64), ("rc", C.c_int32), ("reserved_", C.c_int32), ("stats", Stats)]


def fit_batch(jobs, inflight=0, want_inliers=True):
    """m3d_cloud_fit_batch: jobs = [(cloud, kind, threshold, max_iteration, probability, seed), ...] -> [Fit, ...] in the order
    given.  ONE call into the library (one release of the interpreter's lock): the jobs of a cloud run in order, clouds on
    different lanes (Cloud(..., lane=k)) or devices side by side."""
    n = len(jobs)
    if n == 0:
        return []
    arr = (FitJob * n)()
    bufs = []
    for k, (cloud, kind, thr, it, prob, seed) in enumerate(jobs):
        j = arr[k]
        j.cloud, j.kind, j.threshold, j.max_iteration, j.probability = cloud._h, int(kind), float(thr), int(it), float(prob)
        if seed is not None:
            j.seed, j.has_seed = int(seed) & 0xFFFFFFFFFFFFFFFF, 1
        if want_inliers:
            b = np.empty(max(cloud.n_created, 1), dtype=np.uint64)      # (one list per job: they are all alive at the end)
            bufs.append(b)
            j.inliers = b.ctypes.data
        else:
            bufs.append(None)
    f = lib().m3d_cloud_fit_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    _check(f(C.cast(arr, C.c_void_p), n, int(inflight)))
    out = []
    for k in range(n):
        j = arr[k]
        inl = bufs[k][: j.n_inliers] if want_inliers else np.zeros(0, dtype=np.uint64)
        st = j.stats.asdict()
        st["n_inliers"] = int(j.n_inliers)
        out.append(Fit(int(j.rc), np.array(j.params[: NUM_PARAMS[int(j.kind)]]), inl, st))
    return out


def fit_multi(kind, xyz, devices, normals=None, threshold=0.01, max_iteration=1000, probability=0.9999, seed=None) -> Fit:
    """m3d_fit_multi: one-shot fit from one process driving the devices of `devices`."""
    xyz = _f64(xyz).reshape(-1, 3)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    n = len(xyz)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    params = np.zeros(NUM_PARAMS[kind])
    inl = np.empty(max(n, 1), dtype=np.uint64)
    ni = C.c_size_t(0)
    st = Stats()
    _s, sref = _seed_ref(seed)
    rc = _check(lib().m3d_fit_multi(kind, _p(xyz), _p(nrm), n, threshold, max_iteration, probability,
                                    C.cast(sref, C.c_void_p) if sref else None, _p(dev), len(dev), _p(params), _p(inl),
                                    C.cast(C.byref(ni), C.c_void_p), C.cast(C.byref(st), C.c_void_p)))
    d = st.asdict()
    d["n_inliers"] = int(ni.value)
    return Fit(rc, params, inl[: ni.value].copy(), d)


def registration_ransac_sharded(src, dst, corr_src, corr_dst, comm, threshold=0.01, max_iter=100000,
                                edge_length_threshold=0.9, confidence=0.999, seed=None, device=0):
    """m3d_registration_ransac_sharded (every rank of `comm` calls it with the same inputs)."""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    cs = np.ascontiguousarray(corr_src, dtype=np.uint64)
    cd = np.ascontiguousarray(corr_dst, dtype=np.uint64)
    if len(cs) != len(cd):
        raise ValueError("correspondence lists differ in length")
    T = np.zeros(16)
    st = RegStats()
    _s, sref = _seed_ref(seed)
    _check(lib().m3d_registration_ransac_sharded(_p(src), len(src), _p(dst), len(dst), _p(cs), _p(cd), len(cs),
                                                 threshold, max_iter, edge_length_threshold, confidence,
                                                 C.cast(sref, C.c_void_p) if sref else None, device,
                                                 comm._h if comm is not None else None, _p(T),
                                                 C.cast(C.byref(st), C.c_void_p)))
    return T.reshape(4, 4), st.asdict()


def kabsch(src, dst, scaling=False, device=0):
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    if len(src) != len(dst):
        raise M3DError(ERR_SIZE_MISMATCH, "The number of points pair is not equal.")
    T = np.zeros(16)
    _check(lib().m3d_kabsch(_p(src), _p(dst), len(src), int(bool(scaling)), device, _p(T)))
    return T.reshape(4, 4)


def registration_ransac(src, dst, corr_src, corr_dst, threshold=0.01, max_iter=100000, edge_length_threshold=0.9,
                        confidence=0.999, seed=None, device=0):
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    cs = np.ascontiguousarray(corr_src, dtype=np.uint64)
    cd = np.ascontiguousarray(corr_dst, dtype=np.uint64)
    if len(cs) != len(cd):
        raise ValueError("correspondence lists differ in length")
    T = np.zeros(16)
    st = RegStats()
    _s, sref = _seed_ref(seed)
    _check(lib().m3d_registration_ransac(_p(src), len(src), _p(dst), len(dst), _p(cs), _p(cd), len(cs), threshold,
                                         max_iter, edge_length_threshold, confidence,
                                         C.cast(sref, C.c_void_p) if sref else None, device, _p(T),
                                         C.cast(C.byref(st), C.c_void_p)))
    return T.reshape(4, 4), st.asdict()


def normals_from_map(xyz, w, h, k=5, view_point=(0.0, 0.0, 0.0), device=0, want_ms=False):
    """m3d_normals_from_map: EstimateNormalsFromMap on an organised point map (h*w, 3) -> (h*w, 3) normals."""
    xyz = _f64(xyz).reshape(-1, 3)
    if len(xyz) != w * h:
        raise M3DError(ERR_INVALID_ARG, "The point cloud size is not equal to given point map size.")
    vp = _f64(view_point).reshape(3).copy()
    out = np.empty((w * h, 3))
    ms = C.c_double(0)
    _check(lib().m3d_normals_from_map(_p(xyz), w, h, k, _p(vp), device, _p(out), C.cast(C.byref(ms), C.c_void_p)))
    return (out, ms.value) if want_ms else out


class IcpStats(C.Structure):
    _fields_ = [("fitness", C.c_double), ("inlier_rmse", C.c_double), ("correspondences", C.c_uint64),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("ms_total", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def registration_icp(src, dst, max_correspondence_distance, init=None, max_iteration=30, relative_fitness=1e-6,
                     relative_rmse=1e-6, device=0, want_correspondences=False):
    """open3d.pipelines.registration.registration_icp(source, target, max_correspondence_distance, init,
    TransformationEstimationPointToPoint(), ICPConvergenceCriteria(...)) -> (T, stats[, correspondences])."""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    Ti = _f64(init).reshape(16).copy() if init is not None else None
    T = np.zeros(16)
    st = IcpStats()
    corr = np.zeros(max(len(src), 1), dtype=np.int64) if want_correspondences else None
    _check(lib().m3d_registration_icp(_p(src), len(src), _p(dst), len(dst), max_correspondence_distance, _p(Ti),
                                      max_iteration, relative_fitness, relative_rmse, device, _p(T),
                                      C.cast(C.byref(st), C.c_void_p), _p(corr)))
    if want_correspondences:
        return T.reshape(4, 4), st.asdict(), corr[: len(src)]
    return T.reshape(4, 4), st.asdict()


SEARCH_KNN, SEARCH_RADIUS, SEARCH_HYBRID = 0, 1, 2


def detect_boundary_points(xyz, normals=None, search=SEARCH_HYBRID, radius=0.01, max_nn=30, angle_threshold=90.0,
                           device=0):
    """m3d_detect_boundary_points -> ascending indices (uint64) of the boundary points."""
    xyz = _f64(xyz).reshape(-1, 3)
    nrm = _f64(normals).reshape(-1, 3) if normals is not None else None
    if nrm is not None and len(nrm) != len(xyz):
        raise ValueError("normals and points differ in length")
    out = np.zeros(max(len(xyz), 1), dtype=np.uint64)
    k = C.c_size_t(0)
    _check(lib().m3d_detect_boundary_points(_p(xyz), _p(nrm), len(xyz), search, radius, max_nn, angle_threshold, device,
                                            _p(out), C.cast(C.byref(k), C.c_void_p)))
    return out[: k.value].copy()


def information_matrix(src, dst, max_correspondence_distance, T, device=0):
    """GetInformationMatrixFromPointClouds -> (6 x 6 information matrix, correspondence count)."""
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    Tm = _f64(T).reshape(16).copy()
    info = np.zeros(36)
    nc = C.c_uint64(0)
    _check(lib().m3d_information_matrix(_p(src), len(src), _p(dst), len(dst), max_correspondence_distance, _p(Tm),
                                        device, _p(info), C.cast(C.byref(nc), C.c_void_p)))
    return info.reshape(6, 6), int(nc.value)


class GlobalRegStats(C.Structure):
    """m3d_global_reg_stats"""
    _fields_ = [("n_matches", C.c_uint64), ("n_info_correspondences", C.c_uint64), ("identity_shortcut", C.c_int32),
                ("device", C.c_int32), ("lane", C.c_int32), ("reserved_", C.c_int32), ("ms_match", C.c_double),
                ("ms_ransac", C.c_double), ("ms_info", C.c_double), ("ms_total", C.c_double), ("ransac", RegStats)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("ransac", "reserved_")}
        d["ransac"] = self.ransac.asdict()
        return d


class FragmentPair(C.Structure):
    """m3d_fragment_pair"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n_src", C.c_size_t), ("n_dst", C.c_size_t),
                ("feat_src", C.c_void_p), ("feat_dst", C.c_void_p), ("seed", C.c_uint64), ("has_seed", C.c_int32),
                ("rc", C.c_int32), ("T", C.c_double * 16), ("info", C.c_double * 36), ("stats", GlobalRegStats)]


def _pair_arrays(src, dst, feat_src, feat_dst):
    src = _f64(src).reshape(-1, 3)
    dst = _f64(dst).reshape(-1, 3)
    fs = _f64(feat_src)
    fd = _f64(feat_dst)
    if fs.ndim != 2 or fd.ndim != 2 or fs.shape[1] != fd.shape[1] or len(fs) != len(src) or len(fd) != len(dst):
        raise ValueError("descriptor matrices must be (N, dim), one row per point, equal dim")
    return src, dst, fs, fd


def global_registration(src, dst, feat_src, feat_dst, voxel_size, max_iter=100000, edge_length_threshold=0.9,
                        confidence=0.999, seed=None, device=0, want_stats=False):
    """m3d_global_registration = ReconstructionPipeline::GlobalRegistration with the Ransac method
    (src/pipeline.cpp:790-828): mutual-NN match -> RANSACSolver(1.4 voxel) -> isIdentity shortcut -> information matrix;
    rejected when info(5,5) / min(Ns, Nt) < 0.3.  Returns (success, pose 4x4, information 6x6[, stats])."""
    src, dst, fs, fd = _pair_arrays(src, dst, feat_src, feat_dst)
    T = np.zeros(16)
    info = np.zeros(36)
    st = GlobalRegStats()
    _s, sref = _seed_ref(seed)
    f = lib().m3d_global_registration
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int,
                  C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = _check(f(_p(src), len(src), _p(dst), len(dst), _p(fs), _p(fd), fs.shape[1], float(voxel_size), int(max_iter),
                  float(edge_length_threshold), float(confidence), C.cast(sref, C.c_void_p) if sref else None, device,
                  _p(T), _p(info), C.cast(C.byref(st), C.c_void_p)))
    out = (rc == 1, T.reshape(4, 4), info.reshape(6, 6))
    return out + (st.asdict(),) if want_stats else out


def global_registration_batch(pairs, voxel_size, max_iter=100000, edge_length_threshold=0.9, confidence=0.999, seeds=None,
                              devices=(0,), inflight=0, want_stats=False):
    """m3d_global_registration_batch: pairs = [(src, dst, feat_src, feat_dst), ...] (BuildPoseGraphForScene's loop over
    fragment pairs, src/pipeline.cpp:428-439) -> [(success, pose, information[, stats]), ...] in the order given.
    seeds: one per pair or None."""
    arrs = [_pair_arrays(*p) for p in pairs]
    n = len(arrs)
    if n == 0:
        return []
    dim = arrs[0][2].shape[1]
    if any(a[2].shape[1] != dim for a in arrs):
        raise ValueError("every pair must use descriptors of the same width")
    fp = (FragmentPair * n)()
    for k, (src, dst, fs, fd) in enumerate(arrs):
        fp[k].src, fp[k].dst, fp[k].n_src, fp[k].n_dst = _addr(src), _addr(dst), len(src), len(dst)
        fp[k].feat_src, fp[k].feat_dst = _addr(fs), _addr(fd)
        if seeds is not None and seeds[k] is not None:
            fp[k].seed, fp[k].has_seed = int(seeds[k]) & 0xFFFFFFFFFFFFFFFF, 1
    dev = (C.c_int * len(devices))(*[int(d) for d in devices])
    f = lib().m3d_global_registration_batch
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_int]
    _check(f(C.cast(fp, C.c_void_p), n, dim, float(voxel_size), int(max_iter), float(edge_length_threshold),
             float(confidence), C.cast(dev, C.c_void_p), len(devices), int(inflight)))
    out = []
    for k in range(n):
        r = (fp[k].rc == 1, np.array(fp[k].T).reshape(4, 4), np.array(fp[k].info).reshape(6, 6))
        out.append(r + (fp[k].stats.asdict(),) if want_stats else r)
    return out


class FragmentView(C.Structure):
    """m3d_fragment_view"""
    _fields_ = [("xyz", C.c_void_p), ("feat", C.c_void_p), ("n", C.c_size_t)]


class PairResult(C.Structure):
    """m3d_pair_result"""
    _fields_ = [("s", C.c_int32), ("t", C.c_int32), ("has_seed", C.c_int32), ("rc", C.c_int32), ("seed", C.c_uint64),
                ("T", C.c_double * 16), ("info", C.c_double * 36), ("stats", GlobalRegStats)]


def register_fragment_pairs(fragments, features, pairs, voxel_size, max_iter=100000, edge_length_threshold=0.9,
                            confidence=0.999, seeds=None, devices=(0,), inflight=0, want_stats=False):
    """m3d_register_fragment_pairs: fragments[i] (N_i, 3), features[i] (N_i, dim), pairs = [(s, t), ...] -> [(success, pose,
    information[, stats]), ...]; every fragment is uploaded once per device and stays resident for the call."""
    pts = [_f64(f).reshape(-1, 3) for f in fragments]
    fts = [_f64(f) for f in features]
    if len(pts) != len(fts) or any(f.ndim != 2 or len(f) != len(p) for f, p in zip(fts, pts)):
        raise ValueError("one (N, dim) descriptor matrix per fragment")
    if not pairs:
        return []
    dim = fts[0].shape[1]
    if any(f.shape[1] != dim for f in fts):
        raise ValueError("every fragment must use descriptors of the same width")
    fv = (FragmentView * max(len(pts), 1))()
    for i, (p_, f_) in enumerate(zip(pts, fts)):
        fv[i].xyz, fv[i].feat, fv[i].n = _addr(p_), _addr(f_), len(p_)
    pr = (PairResult * len(pairs))()
    for k, (s_, t_) in enumerate(pairs):
        pr[k].s, pr[k].t = int(s_), int(t_)
        if seeds is not None and seeds[k] is not None:
            pr[k].seed, pr[k].has_seed = int(seeds[k]) & 0xFFFFFFFFFFFFFFFF, 1
    dev = (C.c_int * len(devices))(*[int(d) for d in devices])
    f = lib().m3d_register_fragment_pairs
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_double, C.c_double,
                  C.c_void_p, C.c_int, C.c_int]
    _check(f(C.cast(fv, C.c_void_p), len(pts), dim, C.cast(pr, C.c_void_p), len(pairs), float(voxel_size), int(max_iter),
             float(edge_length_threshold), float(confidence), C.cast(dev, C.c_void_p), len(devices), int(inflight)))
    out = []
    for k in range(len(pairs)):
        r = (pr[k].rc == 1, np.array(pr[k].T).reshape(4, 4), np.array(pr[k].info).reshape(6, 6))
        out.append(r + (pr[k].stats.asdict(),) if want_stats else r)
    return out


class RegSession:
    """m3d_reg: compute_transformation_ransac cut into begin_chunk / validate / replay (multi-GPU driver:
    misc3d_amd.distributed.registration_ransac_sharded).  Every rank must pass the same explicit seed."""

    def __init__(self, src, dst, corr_src, corr_dst, threshold=0.01, max_iter=100000, edge_length_threshold=0.9,
                 confidence=0.999, seed=0, device=0):
        src = _f64(src).reshape(-1, 3)
        dst = _f64(dst).reshape(-1, 3)
        cs = np.ascontiguousarray(corr_src, dtype=np.uint64)
        cd = np.ascontiguousarray(corr_dst, dtype=np.uint64)
        if len(cs) != len(cd):
            raise ValueError("correspondence lists differ in length")
        _s, sref = _seed_ref(seed)
        self._h = lib().m3d_reg_create(_p(src), len(src), _p(dst), len(dst), _p(cs), _p(cd), len(cs), threshold,
                                       max_iter, edge_length_threshold, confidence,
                                       C.cast(sref, C.c_void_p) if sref else None, device)
        if not self._h:
            msg = last_error()
            code = ERR_TOO_FEW_POINTS if "less than 3" in msg else (ERR_DEVICE if "HIP" in msg or "device" in msg
                                                                     else ERR_INVALID_ARG)
            raise M3DError(code, msg)

    def close(self):
        if getattr(self, "_h", None):
            lib().m3d_reg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def begin_chunk(self):
        """-> number of survivors of the next chunk, or None when the loop is over"""
        ns = C.c_size_t(0)
        rc = _check(lib().m3d_reg_begin_chunk(self._h, C.cast(C.byref(ns), C.c_void_p)))
        return int(ns.value) if rc == 1 else None

    def validate(self, s_begin, s_end):
        n = max(s_end - s_begin, 0)
        counts = np.zeros(max(n, 1), dtype=np.uint32)
        sums = np.zeros(max(n, 1), dtype=np.float64)
        _check(lib().m3d_reg_validate(self._h, s_begin, s_end, _p(counts), _p(sums)))
        return counts[:n], sums[:n]

    def replay(self, counts, sums):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        if len(counts) == 0:
            counts, sums = np.zeros(1, dtype=np.uint32), np.zeros(1)
        _check(lib().m3d_reg_replay(self._h, _p(counts), _p(sums)))

    def finish(self):
        T = np.zeros(16)
        st = RegStats()
        _check(lib().m3d_reg_finish(self._h, _p(T), C.cast(C.byref(st), C.c_void_p)))
        return T.reshape(4, 4), st.asdict()


def match_last_fallbacks():
    return int(lib().m3d_match_last_fallbacks())


def match_mutual_nn(feat_src, feat_dst, method=1, n_trees=4, device=0):
    """feat_*: (N, dim) C-contiguous == Eigen dim x N column-major."""
    fs = _f64(feat_src)
    fd = _f64(feat_dst)
    if fs.ndim != 2 or fd.ndim != 2 or fs.shape[1] != fd.shape[1]:
        raise ValueError("descriptor matrices must be (N, dim) with equal dim")
    o0 = np.zeros(max(len(fs), 1), dtype=np.uint64)
    o1 = np.zeros(max(len(fs), 1), dtype=np.uint64)
    k = C.c_size_t(0)
    _check(lib().m3d_match_mutual_nn(_p(fs), len(fs), _p(fd), len(fd), fs.shape[1], method, n_trees, device, _p(o0),
                                     _p(o1), C.cast(C.byref(k), C.c_void_p)))
    return o0[: k.value].copy(), o1[: k.value].copy()


def replay(n_points, kind, max_iteration, probability, valid, counts, rmse=None):
    """Host-only replay of ransac.h:573-575,592-613 over gathered (valid, count) records."""
    st = ReplayState()
    lib().m3d_replay_init(C.byref(st))
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)

    def _cb(_user, i):
        return float(rmse(int(i))) if rmse is not None else 0.0

    cb = RMSE_FN(_cb)
    lib().m3d_replay_chunk(C.byref(st), n_points, kind, max_iteration, probability, 0, len(valid), _p(valid),
                           _p(counts), cb, None)
    return st
