"""ctypes binding of libcup2d_hip.so (include/cup2d_hip.h).  No compute happens in Python."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# enum mirrors (include/cup2d_hip.h)
TMP, CHI, VEL, VOLD, PRES, POLD, TMPV = range(7)
FIELD_DIM = {TMP: 1, CHI: 1, VEL: 2, VOLD: 2, PRES: 1, POLD: 1, TMPV: 2}
MATH_FAST, MATH_STRICT = 0, 1
BLOCKS_ALL, BLOCKS_INNER, BLOCKS_HALO = 0, 1, 2
WALL = -1
(T_ADVECT_STAGE, T_POISSON_RHS, T_SWEEP_A, T_SWEEP_B, T_SWEEP_C, T_SWEEP_D, T_SWEEP_E, T_SCALARS, T_PROJECT,
 T_REDUCE, T_HALO, T_INIT_RESIDUAL) = range(12)
TIMER_NAMES = ['advect_stage', 'poisson_rhs', 'sweep_A', 'sweep_B', 'sweep_C', 'sweep_D', 'sweep_E', 'scalars', 'project',
               'reduce', 'halo', 'init_residual', 'smoother', 'sweep_EA', 'advect_stage2', 'final_x']

EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
WAIT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)

# every symbol include/cup2d_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "cup2d_amr_trace_reads", "cup2d_amr_blocks_reading_ghosts", "cup2d_halo_plan_cells", "cup2d_comm_set_cell_counts",
    "cup2d_create", "cup2d_destroy", "cup2d_last_error", "cup2d_version", "cup2d_set_stream", "cup2d_get_stream",
    "cup2d_synchronize", "cup2d_set_math", "cup2d_upload", "cup2d_download", "cup2d_upload_slab",
    "cup2d_download_slab", "cup2d_field_ptr", "cup2d_fill", "cup2d_copy_field", "cup2d_advect_diffuse_rhs",
    "cup2d_advect_diffuse_rk2", "cup2d_advect_diffuse_stage", "cup2d_vorticity", "cup2d_pressure_rhs",
    "cup2d_laplacian_sub", "cup2d_poisson_rhs", "cup2d_pressure_correction", "cup2d_add_correction", "cup2d_project",
    "cup2d_max_abs_vel", "cup2d_compute_dt", "cup2d_poisson_solve", "cup2d_apply_A", "cup2d_precond",
    "cup2d_get_P_inv", "cup2d_step", "cup2d_halo_plan", "cup2d_halo_pack", "cup2d_halo_unpack",
    "cup2d_halo_pack_vec", "cup2d_halo_unpack_vec", "cup2d_set_comm", "cup2d_set_comm_strip_capacity", "cup2d_set_timing", "cup2d_get_timing", "cup2d_debug_walk_knockout",
    "cup2d_set_P_inv", "cup2d_set_precond", "cup2d_set_matrix_coo", "cup2d_clear_matrix", "cup2d_set_gather", "cup2d_matrix_stats", "cup2d_amr_install_poisson", "cup2d_trim_pool",
    "cup2d_set_solver", "cup2d_set_solver_form", "cup2d_set_nrank_organisation", "cup2d_get_last_solver", "cup2d_get_last_solver_form", "cup2d_get_placement", "cup2d_solver_keep_last", "cup2d_solver_last_iterate", "cup2d_set_amr", "cup2d_amr_poisson_coo", "cup2d_amr_tables", "cup2d_amr_validate_states", "cup2d_amr_regrid", "cup2d_amr_regrid_plan", "cup2d_amr_regrid_changed", "cup2d_amr_regrid_local", "cup2d_amr_regrid_device", "cup2d_amr_regrid_jobs",
    "cup2d_download_blocks", "cup2d_upload_blocks", "cup2d_copy_blocks",
    "cup2d_jacobi_sweeps", "cup2d_poisson_residual", "cup2d_block_linf",
    "cup2d_comm_unique_id", "cup2d_comm_init", "cup2d_comm_finalize", "cup2d_comm_selftest", "cup2d_comm_stats", "cup2d_halo_exchange",
    "cup2d_body_set", "cup2d_body_clear", "cup2d_body_momentum", "cup2d_penalize", "cup2d_amr_set_finest_level",
]
COMM_ID_BYTES = 256
OK, ERR_ARG, ERR_HIP, ERR_NODEVICE, ERR_UNSUPPORTED, ERR_COMM = 0, -1, -2, -3, -4, -5
AMR_WALL, AMR_SAME, AMR_COARSER, AMR_FINER = range(4)
SOLVER_SWEEPS, SOLVER_FUSED = 0, 1
PRECOND_LDS, PRECOND_MFMA, PRECOND_FD = 0, 1, 2


class Cup2dError(RuntimeError):
    pass


def library_path():
    # CUP2D_LIB: another build of the same library (A/B timing of kernel variants on one box)
    return os.environ.get("CUP2D_LIB") or os.path.join(_HERE, "libcup2d_hip.so")


def load_library():
    """Load libcup2d_hip.so; raises Cup2dError if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise Cup2dError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(make -C cup2d_amd/csrc). There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    L.cup2d_last_error.restype = ctypes.c_char_p
    L.cup2d_version.restype = ctypes.c_char_p
    L.cup2d_destroy.restype = None
    d = ctypes.c_double
    vp = ctypes.c_void_p
    i = ctypes.c_int
    L.cup2d_create.argtypes = [ctypes.POINTER(vp), i, i, i, vp, d, i]
    L.cup2d_destroy.argtypes = [vp]
    L.cup2d_set_stream.argtypes = [vp, vp]
    L.cup2d_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.cup2d_synchronize.argtypes = [vp]
    L.cup2d_set_math.argtypes = [vp, i]
    L.cup2d_upload.argtypes = [vp, i, vp]
    L.cup2d_download.argtypes = [vp, i, vp]
    L.cup2d_upload_slab.argtypes = [vp, i, vp]
    L.cup2d_download_slab.argtypes = [vp, i, vp]
    L.cup2d_field_ptr.argtypes = [vp, i, ctypes.POINTER(vp)]
    L.cup2d_fill.argtypes = [vp, i, d]
    L.cup2d_copy_field.argtypes = [vp, i, i]
    L.cup2d_advect_diffuse_rhs.argtypes = [vp, d, d, i]
    L.cup2d_advect_diffuse_rk2.argtypes = [vp, d, d]
    L.cup2d_advect_diffuse_stage.argtypes = [vp, d, d, i, i]
    L.cup2d_vorticity.argtypes = [vp, i]
    L.cup2d_pressure_rhs.argtypes = [vp, d, i, i]
    L.cup2d_laplacian_sub.argtypes = [vp, i]
    L.cup2d_poisson_rhs.argtypes = [vp, d, i]
    L.cup2d_pressure_correction.argtypes = [vp, d, i]
    L.cup2d_add_correction.argtypes = [vp]
    L.cup2d_project.argtypes = [vp, d]
    L.cup2d_max_abs_vel.argtypes = [vp, ctypes.POINTER(d)]
    L.cup2d_compute_dt.argtypes = [vp, d, d, ctypes.POINTER(d)]
    L.cup2d_poisson_solve.argtypes = [vp, d, d, i, i, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(d), ctypes.POINTER(d)]
    L.cup2d_apply_A.argtypes = [vp, i, i]
    L.cup2d_precond.argtypes = [vp, i, i]
    L.cup2d_get_P_inv.argtypes = [vp, vp]
    L.cup2d_set_P_inv.argtypes = [vp, vp]
    L.cup2d_set_precond.argtypes = [vp, i]
    L.cup2d_set_solver.argtypes = [vp, i, i]
    L.cup2d_set_solver_form.argtypes = [vp, i]
    L.cup2d_set_nrank_organisation.argtypes = [vp, i, i]
    L.cup2d_get_last_solver.argtypes = [vp, ctypes.POINTER(i)]
    L.cup2d_get_last_solver_form.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i)]
    L.cup2d_get_placement.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(d), ctypes.POINTER(d), ctypes.POINTER(d)]
    L.cup2d_solver_keep_last.argtypes = [vp, i]
    L.cup2d_solver_last_iterate.argtypes = [vp, i, ctypes.POINTER(d)]
    L.cup2d_set_amr.argtypes = [vp, d, vp, vp, vp, vp]
    L.cup2d_amr_poisson_coo.argtypes = [i, vp, vp, vp, ctypes.c_longlong, vp, vp, vp]
    L.cup2d_amr_poisson_coo.restype = ctypes.c_longlong
    L.cup2d_block_linf.argtypes = [vp, i, vp]
    L.cup2d_jacobi_sweeps.argtypes = [vp, d, i, vp]
    L.cup2d_poisson_residual.argtypes = [vp, vp]
    L.cup2d_amr_tables.argtypes = [i, vp, i, i, vp, vp, vp]
    L.cup2d_amr_validate_states.argtypes = [i, vp, i, i, i, vp]
    L.cup2d_amr_regrid.argtypes = [i, vp, i, i, i, vp, i, vp, vp, vp, ctypes.c_longlong, vp, vp]
    L.cup2d_amr_regrid.restype = ctypes.c_longlong
    L.cup2d_amr_regrid_plan.argtypes = [i, vp, i, i, i, vp, ctypes.c_longlong, vp, vp, vp]
    L.cup2d_amr_regrid_plan.restype = ctypes.c_longlong
    L.cup2d_amr_regrid_changed.argtypes = [i, vp, i, i, i, vp, i, vp, vp, vp, ctypes.c_longlong, vp, vp]
    L.cup2d_amr_regrid_changed.restype = ctypes.c_longlong
    LLc = ctypes.c_longlong
    L.cup2d_amr_regrid_local.argtypes = [i, vp, i, i, i, vp, LLc, LLc, LLc, vp, vp, vp, i, vp, vp, vp, vp, vp]
    L.cup2d_amr_regrid_local.restype = ctypes.c_longlong
    L.cup2d_amr_regrid_device.argtypes = [vp, vp, i, vp, i, i, i, vp, i, vp]
    L.cup2d_amr_regrid_jobs.argtypes = [i, vp, i, i, i, vp, LLc, vp, LLc, vp, vp]
    L.cup2d_amr_regrid_jobs.restype = ctypes.c_longlong
    L.cup2d_download_blocks.argtypes = [vp, i, i, vp, vp]
    L.cup2d_upload_blocks.argtypes = [vp, i, i, vp, vp]
    L.cup2d_copy_blocks.argtypes = [vp, vp, i, i, vp, vp]
    L.cup2d_set_matrix_coo.argtypes = [vp, i, ctypes.c_longlong, vp, vp, vp]
    L.cup2d_clear_matrix.argtypes = [vp]
    L.cup2d_set_gather.argtypes = [vp, i, vp]
    L.cup2d_matrix_stats.argtypes = [vp, vp, vp, vp]
    L.cup2d_amr_install_poisson.argtypes = [vp]
    L.cup2d_trim_pool.argtypes = []
    L.cup2d_step.argtypes = [vp, d, d, d, d, i, i, ctypes.POINTER(d), ctypes.POINTER(i), ctypes.POINTER(d)]
    L.cup2d_halo_plan.argtypes = [vp, i, vp, vp, i, vp, vp]
    L.cup2d_halo_pack.argtypes = [vp, i, i, vp]
    L.cup2d_halo_unpack.argtypes = [vp, i, i, vp]
    L.cup2d_halo_pack_vec.argtypes = [vp, vp, i, i, vp]
    L.cup2d_halo_unpack_vec.argtypes = [vp, vp, i, i, vp]
    L.cup2d_set_comm.argtypes = [vp, EXCHANGE_FN, WAIT_FN, ALLREDUCE_FN, vp, vp, vp, vp]
    L.cup2d_set_comm_strip_capacity.argtypes = [vp, ctypes.c_int]
    L.cup2d_amr_trace_reads.argtypes = [i, vp, vp, vp, i, vp, i, vp]
    L.cup2d_amr_blocks_reading_ghosts.argtypes = [i, i, vp, vp, vp, i, vp]
    L.cup2d_halo_plan_cells.argtypes = [vp, i, i, vp, i, vp]
    L.cup2d_comm_set_cell_counts.argtypes = [vp, i, i, vp, vp, vp, vp]
    L.cup2d_amr_set_finest_level.argtypes = [vp, i]
    L.cup2d_body_set.argtypes = [vp, i, i, vp, vp, vp, vp, d, d]
    L.cup2d_body_clear.argtypes = [vp]
    L.cup2d_body_momentum.argtypes = [vp, i, d, d, vp, vp]
    L.cup2d_penalize.argtypes = [vp, d, d, vp]
    L.cup2d_comm_unique_id.argtypes = [vp]
    L.cup2d_comm_init.argtypes = [vp, i, i, vp, i, vp, vp, vp, vp, vp]
    L.cup2d_comm_finalize.argtypes = [vp]
    L.cup2d_comm_selftest.argtypes = [vp, d, ctypes.c_char_p, i]
    LL = ctypes.POINTER(ctypes.c_longlong)
    L.cup2d_comm_stats.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i), LL, LL, LL]
    L.cup2d_halo_exchange.argtypes = [vp, i, i]
    L.cup2d_set_timing.argtypes = [vp, i]
    L.cup2d_debug_walk_knockout.argtypes = [vp, i]
    L.cup2d_get_timing.argtypes = [vp, i, ctypes.POINTER(d), ctypes.POINTER(i)]
    _LIB = L
    return L


CELLS_HALO1, CELLS_HALO3, CELLS_MATRIX = 0, 1, 2  # cup2d_halo_plan_cells sets
CELL_SET_NAMES = ("halo1", "halo3", "matrix")


def cell_strip(strip_doubles):
    """(set, doubles per cell) of a negative strip_doubles the exchange callback receives (CUP2D_CELL_STRIP), else None"""
    if strip_doubles >= 0:
        return None
    return (-strip_doubles) >> 4, (-strip_doubles) & 15


def check(status, what=""):
    if status != 0:
        L = load_library()
        raise Cup2dError("%s failed with status %d: %s" % (what or "cup2d call", status, L.cup2d_last_error().decode()))
