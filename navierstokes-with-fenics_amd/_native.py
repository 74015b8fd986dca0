"""ctypes binding of libnsfem_hip.so (C ABI: include/nsfem.h).

This is the only way the Python host code reaches the device.  There is NO CPU
fallback: if the shared library is missing or no MI355X is visible, every entry
point raises.  (The reference reaches its native layer -- DOLFIN/PETSc -- through
pybind11 inside ``import dolfin``; this thin ctypes layer replaces that import.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnsfem_hip.so")

# ---- enums (mirror include/nsfem.h) ------------------------------------------
OK, ERR_ARG, ERR_HIP, ERR_BREAKDOWN, ERR_NOT_CONVERGED, ERR_COMM = 0, -1, -2, -3, -4, -5
U0, U1, U2, USTAR, P, P_OLD, BODY_FORCE, TRACTION, P2_OLD = range(9)
VELOCITY, PRESSURE, PRESSURE_PRECOND = 0, 1, 2
(OP_MASS_P2, OP_STIFF_P2, OP_STIFF_P1, OP_MASS_P1, OP_DIV, OP_GRAD, OP_DIVT,
 OP_MOMENTUM_JAC, OP_VISCOUS_EXTRA, OP_MOMENTUM_JAC_MF, OP_MOMENTUM_SMOOTHER,
 OP_CONVECTION_ACTION) = range(12)
SYS_MOMENTUM, SYS_POISSON, SYS_CORRECTION, SYS_MONOLITHIC = range(4)
MAX_NEWTON = 64

EXPORTED_SYMBOLS = (
    "nsfem_create", "nsfem_destroy", "nsfem_last_error", "nsfem_version",
    "nsfem_set_coeffs", "nsfem_set_bdf", "nsfem_set_dirichlet", "nsfem_set_viscous_form",
    "nsfem_set_convective_form",
    "nsfem_set_state", "nsfem_get_state", "nsfem_state_size", "nsfem_state_devptr",
    "nsfem_assemble", "nsfem_residual_norm", "nsfem_get_rhs", "nsfem_solve",
    "nsfem_operator_shape", "nsfem_operator_export", "nsfem_operator_apply", "nsfem_kernel_apply",
    "nsfem_default_step_opts", "nsfem_step_ipcs", "nsfem_step_bdf", "nsfem_advance",
    "nsfem_shift_mean_pressure", "nsfem_time_spmv", "nsfem_synchronize", "nsfem_mass_solve",
    "nsfem_mg_add_level", "nsfem_mg_finalize", "nsfem_mg_set_global_coarse", "nsfem_mg_set_global_coarse_constrained",
    "nsfem_mg_set_schur_operator", "nsfem_mg_set_schur_mode", "nsfem_set_halo_lists", "nsfem_smoother_info", "nsfem_mg_set_global_index", "nsfem_comm_allreduce", "nsfem_mg_add_global_level", "nsfem_cfl_number", "nsfem_set_angular_velocity", "nsfem_set_angular_velocity_3d", "nsfem_profile_smoother", "nsfem_profile_smoother_detail", "nsfem_profile_convection", "nsfem_jacobian_info", "nsfem_set_preconditioner_shift", "nsfem_poisson_solve", "nsfem_p2_mass_bounds", "nsfem_mg_set_truncation", "nsfem_comm_stats", "nsfem_mg_set_halo_mode", "nsfem_set_overlap", "nsfem_comm_overlapped", "nsfem_boundary_force",
    "nsfem_set_partition", "nsfem_comm_unique_id", "nsfem_comm_attach_rccl",
    "nsfem_comm_local_create", "nsfem_comm_local_destroy", "nsfem_comm_attach_local", "nsfem_comm_attach_shm",
    "nsfem_mg_apply", "nsfem_mg_info", "nsfem_poisson_set_fast_diag", "nsfem_poisson_set_fast_diag_rows",
    "nsfem_operator_diagonal",
)


class MeshDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_cells", C.c_int32), ("n_vertices", C.c_int32),
                ("n_p2", C.c_int32), ("n_p1", C.c_int32),
                ("coords", C.POINTER(C.c_double)), ("cells", C.POINTER(C.c_int32)),
                ("p2_dofmap", C.POINTER(C.c_int32)), ("p1_dofmap", C.POINTER(C.c_int32))]


class KernelTest(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("space", "nv", "family", "epilogue", "steps", "maskmode", "ghost", "ident",
                                          "from_zero", "with_residual", "dict_ok", "used_family", "dict_entries",
                                          "dict_exact", "lattice_w", "reserved")] + \
               [("a", C.c_double), ("b_coef", C.c_double), ("c1", C.c_double * 8), ("c2", C.c_double * 8),
                ("x", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double)), ("d", C.POINTER(C.c_double)),
                ("mask", C.POINTER(C.c_uint8)),
                ("y", C.POINTER(C.c_double)), ("d_out", C.POINTER(C.c_double)), ("r_out", C.POINTER(C.c_double))]


class KrylovOpts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("max_iter", C.c_int32),
                ("precond", C.c_int32), ("check_every", C.c_int32), ("first_check", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("residual", C.c_double), ("residual0", C.c_double)]


class StepOpts(C.Structure):
    _fields_ = [("newton_atol", C.c_double), ("newton_rtol", C.c_double),
                ("newton_max_iter", C.c_int32), ("convective_form", C.c_int32),
                ("momentum", KrylovOpts), ("poisson", KrylovOpts), ("correction", KrylovOpts),
                ("picard", C.c_int32), ("allow_nonconvergence", C.c_int32),
                ("newton_forcing", C.c_double), ("matrix_free", C.c_int32),
                ("pressure_extrapolation", C.c_int32)]


class StepInfo(C.Structure):
    _fields_ = [("newton_iterations", C.c_int32), ("krylov_iterations_momentum", C.c_int32),
                ("krylov_iterations_poisson", C.c_int32),
                ("krylov_iterations_correction", C.c_int32),
                ("newton_residuals", C.c_double * MAX_NEWTON),
                ("converged", C.c_int32), ("reserved", C.c_int32)]


class Halo(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("send_up_off", "send_up_cnt", "recv_above_off",
                                         "recv_above_cnt", "send_down_off", "send_down_cnt",
                                         "recv_below_off", "recv_below_cnt")]

    @classmethod
    def from_dict(cls, d):
        h = cls()
        if d:
            for key in ("send_up", "recv_above", "send_down", "recv_below"):
                setattr(h, key + "_off", int(d[key][0]))
                setattr(h, key + "_cnt", int(d[key][1]))
        return h


class HaloLists(C.Structure):
    _fields_ = [("n_neighbours", C.c_int32), ("neighbour", C.POINTER(C.c_int32)),
                ("send_ptr", C.POINTER(C.c_int64)), ("send_idx", C.POINTER(C.c_int32)),
                ("recv_ptr", C.POINTER(C.c_int64)), ("recv_idx", C.POINTER(C.c_int32))]


class MgLevelDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_cells", C.c_int32),
                ("coords", C.POINTER(C.c_double)), ("cells", C.POINTER(C.c_int32)),
                ("n_fine", C.c_int32), ("p_rowptr", C.POINTER(C.c_int32)),
                ("p_col", C.POINTER(C.c_int32)), ("p_val", C.POINTER(C.c_double)),
                ("ghost", C.POINTER(C.c_uint8)), ("halo", Halo),
                ("dofmap", C.POINTER(C.c_int32)), ("n_dofs", C.c_int32), ("transfer_kind", C.c_int32)]


class PartitionDesc(C.Structure):
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32),
                ("p2_ghost", C.POINTER(C.c_uint8)), ("p1_ghost", C.POINTER(C.c_uint8)),
                ("p2_halo", Halo), ("p1_halo", Halo),
                ("n_p2_global", C.c_int64), ("n_p1_global", C.c_int64), ("periodic", C.c_int32)]


class MgOpts(C.Structure):
    _fields_ = [("smoother_degree", C.c_int32), ("coarse_dense_max", C.c_int32),
                ("eig_ratio", C.c_double)]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libnsfem_hip: %s (status %d)" % (message, code))
        self.code = code


_lib = None


def load_library(path=None):
    """dlopen the shared library and declare the prototypes.  Raises loudly when
    the library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("libnsfem_hip.so is not built (%s): the HIP extension is mandatory, "
                          "there is no CPU fallback; run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    protos = {
        "nsfem_create": (C.c_int, [C.POINTER(MeshDesc), C.c_int, C.POINTER(vp)]),
        "nsfem_destroy": (None, [vp]),
        "nsfem_last_error": (C.c_char_p, [vp]),
        "nsfem_version": (C.c_int, []),
        "nsfem_set_coeffs": (C.c_int, [vp, pd]),
        "nsfem_set_bdf": (C.c_int, [vp, pd, dbl]),
        "nsfem_set_dirichlet": (C.c_int, [vp, C.c_int, i32, pi, pd]),
        "nsfem_set_viscous_form": (C.c_int, [vp, C.c_int]),
        "nsfem_set_convective_form": (C.c_int, [vp, C.c_int, C.c_int]),
        "nsfem_set_state": (C.c_int, [vp, C.c_int, pd, i64]),
        "nsfem_get_state": (C.c_int, [vp, C.c_int, pd, i64]),
        "nsfem_state_size": (i64, [vp, C.c_int]),
        "nsfem_state_devptr": (vp, [vp, C.c_int]),
        "nsfem_assemble": (C.c_int, [vp, C.c_int, C.c_uint32]),
        "nsfem_residual_norm": (C.c_int, [vp, C.c_int, pd]),
        "nsfem_get_rhs": (C.c_int, [vp, C.c_int, pd, i64]),
        "nsfem_solve": (C.c_int, [vp, C.c_int, C.POINTER(KrylovOpts), C.POINTER(SolveInfo)]),
        "nsfem_operator_shape": (C.c_int, [vp, C.c_int, C.POINTER(i64), C.POINTER(i64),
                                           C.POINTER(i64)]),
        "nsfem_operator_export": (C.c_int, [vp, C.c_int, pi, pi, pd]),
        "nsfem_operator_apply": (C.c_int, [vp, C.c_int, pd, pd]),
        "nsfem_kernel_apply": (C.c_int, [vp, C.POINTER(KernelTest)]),
        "nsfem_default_step_opts": (C.c_int, [C.POINTER(StepOpts)]),
        "nsfem_step_ipcs": (C.c_int, [vp, C.POINTER(StepOpts), C.POINTER(StepInfo)]),
        "nsfem_step_bdf": (C.c_int, [vp, C.POINTER(StepOpts), C.POINTER(StepInfo)]),
        "nsfem_advance": (C.c_int, [vp, C.c_int]),
        "nsfem_shift_mean_pressure": (C.c_int, [vp, dbl, pd]),
        "nsfem_cfl_number": (C.c_int, [vp, C.c_int, dbl, pd]),
        "nsfem_set_angular_velocity": (C.c_int, [vp, dbl, dbl]),
        "nsfem_set_angular_velocity_3d": (C.c_int, [vp, pd, pd]),
        "nsfem_set_preconditioner_shift": (C.c_int, [vp, dbl]),
        "nsfem_p2_mass_bounds": (C.c_int, [C.c_int, pd, pd]),
        "nsfem_mg_set_truncation": (C.c_int, [vp, dbl, dbl]),
        "nsfem_comm_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.c_int]),
        "nsfem_mg_set_halo_mode": (C.c_int, [vp, C.c_int]),
        "nsfem_set_overlap": (C.c_int, [vp, C.c_int]),
        "nsfem_boundary_force": (C.c_int, [vp, C.c_int, C.c_int, i32, pi, pi, dbl, dbl, pd]),
        "nsfem_comm_overlapped": (C.c_int, [vp, C.POINTER(C.c_int64), C.c_int]),
        "nsfem_poisson_solve": (C.c_int, [vp, pd, i64, pi, pd, C.POINTER(KrylovOpts), C.POINTER(SolveInfo)]),
        "nsfem_profile_smoother": (C.c_int, [vp, C.c_int, pd, C.POINTER(i64), C.POINTER(i64)]),
        "nsfem_profile_convection": (C.c_int, [vp, C.c_int, pd, C.POINTER(i64), C.POINTER(i64)]),
        "nsfem_profile_smoother_detail": (C.c_int, [vp, C.POINTER(i64)]),
        "nsfem_time_spmv": (C.c_int, [vp, C.c_int, C.c_int, pd, C.POINTER(i64)]),
        "nsfem_synchronize": (C.c_int, [vp]),
        "nsfem_mg_add_level": (C.c_int, [vp, C.POINTER(MgLevelDesc)]),
        "nsfem_mg_finalize": (C.c_int, [vp, C.POINTER(MgOpts)]),
        "nsfem_mg_add_global_level": (C.c_int, [vp, C.POINTER(MgLevelDesc)]),
        "nsfem_mg_set_schur_operator": (C.c_int, [vp, C.c_int, C.c_int32, C.POINTER(C.c_int32),
                                                  C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                  C.c_int]),
        "nsfem_mg_set_schur_mode": (C.c_int, [vp, C.c_int]),
        "nsfem_smoother_info": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "nsfem_jacobian_info": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "nsfem_mg_apply": (C.c_int, [vp, C.c_int, pd, pd]),
        "nsfem_mg_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64)]),
        "nsfem_poisson_set_fast_diag": (C.c_int, [vp, i32, i32, pd, pd, pd]),
        "nsfem_operator_diagonal": (C.c_int, [vp, C.c_int, pd]),
        "nsfem_poisson_set_fast_diag_rows": (C.c_int, [vp, i32, i32, i32, pd, pd, pd]),
        "nsfem_set_halo_lists": (C.c_int, [vp, C.c_int, C.POINTER(HaloLists)]),
        "nsfem_mg_set_global_index": (C.c_int, [vp, i32, pi]),
        "nsfem_comm_allreduce": (C.c_int, [vp, pd, C.c_int, C.c_int]),
        "nsfem_mg_set_global_coarse": (C.c_int, [vp, i32, i32, pd, pi, i64]),
        "nsfem_mg_set_global_coarse_constrained": (C.c_int, [vp, i32, i32, pd, pi, pi, i32, i64]),
        "nsfem_set_partition": (C.c_int, [vp, C.POINTER(PartitionDesc)]),
        "nsfem_comm_unique_id": (C.c_int, [C.c_char_p]),
        "nsfem_comm_attach_rccl": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
        "nsfem_comm_local_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "nsfem_comm_local_destroy": (None, [vp]),
        "nsfem_comm_attach_local": (C.c_int, [vp, vp, C.c_int]),
        "nsfem_comm_attach_shm": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, C.c_int64]),
        "nsfem_mass_solve": (C.c_int, [vp, C.c_int, pd, pd, C.POINTER(KrylovOpts),
                                       C.POINTER(SolveInfo)]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)      # AttributeError = missing export: fail loudly
        fn.restype = res
        fn.argtypes = args
    if path == LIB_PATH:
        _lib = lib
    return lib


def local_group_create(size):
    """In-process communicator group (several contexts on one device; testing only)."""
    lib = load_library()
    g = C.c_void_p()
    rc = lib.nsfem_comm_local_create(int(size), C.byref(g))
    if rc != OK:
        raise NativeError(rc, "nsfem_comm_local_create failed")
    return g


def local_group_destroy(group):
    load_library().nsfem_comm_local_destroy(group)


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    rc = load_library().nsfem_comm_unique_id(buf)
    if rc != OK:
        raise NativeError(rc, "ncclGetUniqueId failed")
    return buf.raw


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class NsfemContext:
    """RAII wrapper of one ``nsfem_ctx`` (one mesh, one GPU, one stream)."""

    def __init__(self, coords, cells, p2_dofmap, p1_dofmap, n_p2, n_p1, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        p2 = np.ascontiguousarray(p2_dofmap, dtype=np.int32)
        p1 = np.ascontiguousarray(p1_dofmap, dtype=np.int32)
        assert coords.ndim == 2 and coords.shape[1] in (2, 3)
        dim = int(coords.shape[1])                        # 2: triangles, 3: tetrahedra
        assert cells.shape == (p2.shape[0], dim + 1) and p1.shape == cells.shape
        assert p2.shape[1] == (6 if dim == 2 else 10)
        self.dim = dim
        desc = MeshDesc(dim, cells.shape[0], coords.shape[0], int(n_p2), int(n_p1),
                        _dp(coords), _ip(cells), _ip(p2), _ip(p1))
        rc = self._lib.nsfem_create(C.byref(desc), int(device), C.byref(self._h))
        if rc != OK:
            msg = self._lib.nsfem_last_error(None)
            raise NativeError(rc, msg.decode() if msg else "nsfem_create failed")
        self.n_p2, self.n_p1 = int(n_p2), int(n_p1)
        self.n_velocity = dim * self.n_p2

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nsfem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            msg = self._lib.nsfem_last_error(self._h)
            raise NativeError(rc, msg.decode() if msg else "error")

    # -- coefficients / BCs ---------------------------------------------------
    def set_coeffs(self, convective, pressure, viscous, body_force=None, coriolis=None, euler=None):
        c = np.array([np.nan if v is None else float(v)
                      for v in (convective, pressure, viscous, body_force, coriolis, euler)])
        self._check(self._lib.nsfem_set_coeffs(self._h, _dp(c)))

    def set_bdf(self, alpha, k):
        a = np.ascontiguousarray(alpha, dtype=np.float64)
        assert a.shape == (3,)
        self._check(self._lib.nsfem_set_bdf(self._h, _dp(a), float(k)))

    def set_dirichlet(self, field, dofs, vals):
        d = np.ascontiguousarray(dofs, dtype=np.int32)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        assert d.shape == v.shape and d.ndim == 1
        self._check(self._lib.nsfem_set_dirichlet(self._h, field, d.size, _ip(d), _dp(v)))

    def set_convective_form(self, form, picard=False):
        self._check(self._lib.nsfem_set_convective_form(self._h, int(form), int(bool(picard))))

    def set_viscous_form(self, traction_form):
        self._check(self._lib.nsfem_set_viscous_form(self._h, int(bool(traction_form))))

    # -- state ------------------------------------------------------------------
    def state_size(self, slot):
        return int(self._lib.nsfem_state_size(self._h, slot))

    def set_state(self, slot, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._lib.nsfem_set_state(self._h, slot, _dp(v), v.size))

    def get_state(self, slot):
        out = np.empty(self.state_size(slot), dtype=np.float64)
        self._check(self._lib.nsfem_get_state(self._h, slot, _dp(out), out.size))
        return out

    def state_devptr(self, slot):
        return self._lib.nsfem_state_devptr(self._h, slot)

    # -- assembly seam + solves -------------------------------------------------
    def assemble(self, system, new_step=False):
        self._check(self._lib.nsfem_assemble(self._h, system, 1 if new_step else 0))

    def residual_norm(self, system):
        out = C.c_double()
        self._check(self._lib.nsfem_residual_norm(self._h, system, C.byref(out)))
        return out.value

    def get_rhs(self, system):
        n = self.n_p1 if system == SYS_POISSON else self.n_velocity
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.nsfem_get_rhs(self._h, system, _dp(out), n))
        return out

    def solve(self, system, rtol=1e-12, atol=1e-14, max_iter=20000, precond=0, check_every=1):
        o = KrylovOpts(rtol, atol, max_iter, precond, check_every, 0)
        info = SolveInfo()
        self._check(self._lib.nsfem_solve(self._h, system, C.byref(o), C.byref(info)))
        return info

    def default_step_opts(self):
        o = StepOpts()
        self._lib.nsfem_default_step_opts(C.byref(o))
        return o

    def step_ipcs(self, opts=None):
        o = opts or self.default_step_opts()
        info = StepInfo()
        self._check(self._lib.nsfem_step_ipcs(self._h, C.byref(o), C.byref(info)))
        return info

    def step_bdf(self, opts=None):
        o = opts or self.default_step_opts()
        info = StepInfo()
        self._check(self._lib.nsfem_step_bdf(self._h, C.byref(o), C.byref(info)))
        return info

    def comm_stats(self, reset=False):
        """{allreduce_calls, allreduce_bytes, exchanges, exchange_bytes} of this rank"""
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_comm_stats(self._h, out, 1 if reset else 0))
        return dict(zip(("allreduce_calls", "allreduce_bytes", "exchanges", "exchange_bytes"), [int(v) for v in out]))

    def set_overlap(self, enable):
        """halo exchanges on the communicator's stream under the interior rows of the products"""
        self._check(self._lib.nsfem_set_overlap(self._h, 1 if enable else 0))

    def comm_overlapped(self, reset=False):
        out = C.c_int64()
        self._check(self._lib.nsfem_comm_overlapped(self._h, C.byref(out), 1 if reset else 0))
        return int(out.value)

    def mg_set_halo_mode(self, relaxed):
        """partitioned multigrid cycles: False / "exact" one halo exchange per product (the serial cycle),
        True / "relaxed" one per smoothing sequence (frozen ghosts in between)"""
        if isinstance(relaxed, str):
            relaxed = {"exact": False, "relaxed": True}[relaxed]
        self._check(self._lib.nsfem_mg_set_halo_mode(self._h, 1 if relaxed else 0))

    def mg_set_truncation(self, max_ratio, coarse_tol=0.1):
        self._check(self._lib.nsfem_mg_set_truncation(self._h, float(max_ratio), float(coarse_tol)))

    def advance(self, scheme=0):
        self._check(self._lib.nsfem_advance(self._h, scheme))

    def shift_mean_pressure(self, target):
        out = C.c_double()
        self._check(self._lib.nsfem_shift_mean_pressure(self._h, float(target), C.byref(out)))
        return out.value

    def mass_solve(self, field, b, rtol=1e-13, max_iter=2000):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty_like(b)
        o = KrylovOpts(rtol, 0.0, max_iter, 0, 1, 0)
        info = SolveInfo()
        self._check(self._lib.nsfem_mass_solve(self._h, field, _dp(b), _dp(x), C.byref(o),
                                               C.byref(info)))
        return x

    def poisson_solve(self, rhs, dirichlet_dofs, rtol=1e-12, max_iter=20000):
        """post-processing P1 Poisson solve with homogeneous Dirichlet dofs (may be empty)"""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        assert rhs.size == self.n_p1
        d = np.ascontiguousarray(dirichlet_dofs, dtype=np.int32)
        x = np.empty(self.n_p1, dtype=np.float64)
        o = KrylovOpts(rtol, 1e-300, max_iter, 0, 1, 0)
        info = SolveInfo()
        self._check(self._lib.nsfem_poisson_solve(self._h, _dp(rhs), d.size, _ip(d), _dp(x), C.byref(o),
                                                  C.byref(info)))
        return x

    def mg_add_level(self, coords, cells, p_rowptr, p_col, p_val, ghost=None, halo=None, dofmap=None, nested=None):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        rp = np.ascontiguousarray(p_rowptr, dtype=np.int32)
        pc = np.ascontiguousarray(p_col, dtype=np.int32)
        pv = np.ascontiguousarray(p_val, dtype=np.float64)
        g = None if ghost is None else np.ascontiguousarray(ghost, dtype=np.uint8)
        gp = g.ctypes.data_as(C.POINTER(C.c_uint8)) if g is not None else None
        dm = None if dofmap is None else np.ascontiguousarray(dofmap, dtype=np.int32)
        assert dm is None or dm.shape == cells.shape
        d = MgLevelDesc(coords.shape[0], cells.shape[0], _dp(coords), _ip(cells), rp.size - 1,
                        _ip(rp), _ip(pc), _dp(pv), gp, Halo.from_dict(halo),
                        _ip(dm) if dm is not None else None, int(dm.max()) + 1 if dm is not None else 0,
                        0 if nested is None else (1 if nested else 2))
        self._check(self._lib.nsfem_mg_add_level(self._h, C.byref(d)))

    def mg_set_schur_operator(self, level, csr, singular):
        """Level ``level`` of the algebraic Schur-complement Laplacian (scipy CSR, sorted)."""
        csr = csr.tocsr()
        csr.sort_indices()
        rp = np.ascontiguousarray(csr.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(csr.indices, dtype=np.int32)
        cv = np.ascontiguousarray(csr.data, dtype=np.float64)
        self._check(self._lib.nsfem_mg_set_schur_operator(self._h, int(level), csr.shape[0],
                                                          _ip(rp), _ip(ci), _dp(cv),
                                                          1 if singular else 0))

    def smoother_info(self):
        """finest-level smoothing kernel of the velocity multigrid: dict(kind, stencils, longest_row,
        csr_bytes) -- kind "csr-stream" | "sell-64" | "stencil-dictionary" """
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_smoother_info(self._h, out))
        return dict(kind=("csr-stream", "sell-64", "stencil-dictionary", "stencil-dictionary")[int(out[0])],
                    multistep_lattice_kernel=int(out[0]) == 3, stencils=int(out[1]),
                    longest_row=abs(int(out[2])), bitwise_exact=int(out[2]) < 0, csr_bytes=int(out[3]))

    def mg_apply(self, which, r):
        """one cycle z = M^-1 r of the pressure (which=0) or velocity (which=1) multigrid preconditioner"""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        self._check(self._lib.nsfem_mg_apply(self._h, int(which), _dp(r), _dp(z)))
        return z

    def poisson_set_fast_diag(self, factors, first_line=None):
        """factors of poisson_fd.factors(): the projection step may then run with Krylov option precond = 3.
        first_line (partitioned strips): the factors belong to the GLOBAL lattice, this context's pressure space is
        its lines first_line, first_line + 1, ... (ghost lines included)"""
        inv = np.ascontiguousarray(factors["inv"], dtype=np.float64)
        vx = np.ascontiguousarray(factors["Vx"], dtype=np.float64)
        vy = np.ascontiguousarray(factors["Vy"], dtype=np.float64)
        H, W = inv.shape
        assert vx.shape == (W, W) and vy.shape == (H, H)
        if first_line is None:
            self._check(self._lib.nsfem_poisson_set_fast_diag(self._h, W, H, _dp(vx), _dp(vy), _dp(inv)))
        else:
            self._check(self._lib.nsfem_poisson_set_fast_diag_rows(self._h, W, H, int(first_line), _dp(vx), _dp(vy),
                                                                   _dp(inv)))

    def mg_info(self, which):
        """dict(legs, launches_per_cycle, levels, leg_launches): how the cycles of a hierarchy run"""
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_mg_info(self._h, int(which), out))
        return dict(legs=int(out[0]), launches_per_cycle=int(out[1]), levels=int(out[2]), leg_launches=int(out[3]))

    def mg_lattice_info(self, which):
        """dict(lattice_levels, lattice_launches, levels, ghost_lines): the multi-step lattice kernel on the Poisson
        (which = 0) / velocity (1) hierarchy; on partitioned strips it runs in relaxed halo mode only"""
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_mg_info(self._h, 2 + int(which), out))
        return dict(lattice_levels=int(out[0]), lattice_launches=int(out[1]), levels=int(out[2]),
                    ghost_lines=(int(out[3]) // 256, int(out[3]) % 256))

    def mg_set_schur_mode(self, additive):
        """partitioned meshes: the Schur operators set afterwards are this rank's additive parts"""
        self._check(self._lib.nsfem_mg_set_schur_mode(self._h, 1 if additive else 0))

    def comm_allreduce(self, values, op="sum"):
        """sum / max over the ranks of a few host doubles (a copy; single contexts: unchanged)"""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self._lib.nsfem_comm_allreduce(self._h, _dp(v), int(v.size), 1 if op == "max" else 0))
        return v

    def mg_add_global_level(self, coords, cells, p_rowptr, p_col, p_val, dofmap=None):
        """coarser level of the replicated hierarchy below the global coarsest mesh"""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        rp = np.ascontiguousarray(p_rowptr, dtype=np.int32)
        pc = np.ascontiguousarray(p_col, dtype=np.int32)
        pv = np.ascontiguousarray(p_val, dtype=np.float64)
        d = MgLevelDesc(coords.shape[0], cells.shape[0], _dp(coords), _ip(cells), rp.size - 1,
                        _ip(rp), _ip(pc), _dp(pv), None, Halo.from_dict(None), None, 0)
        if dofmap is not None:
            dm = np.ascontiguousarray(dofmap, dtype=np.int32)
            assert dm.shape == cells.shape
            d.dofmap, d.n_dofs = _ip(dm), int(dm.max()) + 1
        self._check(self._lib.nsfem_mg_add_global_level(self._h, C.byref(d)))

    def mg_set_global_coarse(self, coords, cells, offset, dofmap=None):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        if dofmap is not None:                     # constrained (periodic) global coarse space
            dm = np.ascontiguousarray(dofmap, dtype=np.int32)
            assert dm.shape == cells.shape
            self._check(self._lib.nsfem_mg_set_global_coarse_constrained(
                self._h, coords.shape[0], cells.shape[0], _dp(coords), _ip(cells), _ip(dm),
                int(dm.max()) + 1, int(offset)))
            return
        self._check(self._lib.nsfem_mg_set_global_coarse(self._h, coords.shape[0], cells.shape[0],
                                                         _dp(coords), _ip(cells), int(offset)))

    # -- multi-GPU --------------------------------------------------------------------
    def set_partition(self, rank, size, p2_ghost, p1_ghost, p2_halo, p1_halo, n_p2_global,
                      n_p1_global, periodic=False):
        g2 = np.ascontiguousarray(p2_ghost, dtype=np.uint8)
        g1 = np.ascontiguousarray(p1_ghost, dtype=np.uint8)
        assert g2.size == self.n_p2 and g1.size == self.n_p1
        d = PartitionDesc(rank, size, g2.ctypes.data_as(C.POINTER(C.c_uint8)),
                          g1.ctypes.data_as(C.POINTER(C.c_uint8)), Halo.from_dict(p2_halo),
                          Halo.from_dict(p1_halo), int(n_p2_global), int(n_p1_global), 1 if periodic else 0)
        self._check(self._lib.nsfem_set_partition(self._h, C.byref(d)))

    def set_halo_lists(self, target, lists):
        """index-list halo of an unstructured partition; ``lists`` = dict(neighbour, send_ptr,
        send_idx, recv_ptr, recv_idx); target 0 P2 nodes, 1 P1 nodes, 2 + l multigrid level l"""
        nb = np.ascontiguousarray(lists["neighbour"], dtype=np.int32)
        sp = np.ascontiguousarray(lists["send_ptr"], dtype=np.int64)
        si = np.ascontiguousarray(lists["send_idx"], dtype=np.int32)
        rp = np.ascontiguousarray(lists["recv_ptr"], dtype=np.int64)
        ri = np.ascontiguousarray(lists["recv_idx"], dtype=np.int32)
        assert sp.size == nb.size + 1 and rp.size == nb.size + 1 and sp[-1] == si.size and rp[-1] == ri.size
        p64 = C.POINTER(C.c_int64)
        d = HaloLists(int(nb.size), _ip(nb), sp.ctypes.data_as(p64), _ip(si), rp.ctypes.data_as(p64), _ip(ri))
        self._check(self._lib.nsfem_set_halo_lists(self._h, int(target), C.byref(d)))

    def mg_set_global_index(self, local_to_global):
        idx = np.ascontiguousarray(local_to_global, dtype=np.int32)
        self._check(self._lib.nsfem_mg_set_global_index(self._h, int(idx.size), _ip(idx)))

    def attach_local_comm(self, group, rank):
        self._check(self._lib.nsfem_comm_attach_local(self._h, group, rank))

    def attach_shm_comm(self, name, rank, size, slot_bytes=0):
        """one process per rank on a SHARED device: host-staged shared-memory communicator"""
        self._check(self._lib.nsfem_comm_attach_shm(self._h, name.encode(), rank, size, int(slot_bytes)))

    def attach_rccl_comm(self, unique_id, rank, size):
        assert len(unique_id) == 128
        self._check(self._lib.nsfem_comm_attach_rccl(self._h, unique_id, rank, size))

    def mg_finalize(self, degree=2, eig_ratio=4.0, coarse_dense_max=1200):
        o = MgOpts(int(degree), int(coarse_dense_max), float(eig_ratio))
        self._check(self._lib.nsfem_mg_finalize(self._h, C.byref(o)))

    def synchronize(self):
        self._check(self._lib.nsfem_synchronize(self._h))

    # -- operator introspection ---------------------------------------------------
    def operator_csr(self, op):
        """scipy CSR copy of a device operator (parity tests)."""
        import scipy.sparse as sp
        nr, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_operator_shape(self._h, op, C.byref(nr), C.byref(ncol),
                                                   C.byref(nnz)))
        rowptr = np.empty(nr.value + 1, dtype=np.int32)
        col = np.empty(nnz.value, dtype=np.int32)
        val = np.empty(nnz.value, dtype=np.float64)
        self._check(self._lib.nsfem_operator_export(self._h, op, _ip(rowptr), _ip(col), _dp(val)))
        return sp.csr_matrix((val, col, rowptr), shape=(nr.value, ncol.value))

    def operator_diagonal(self, op):
        """diagonal of a square scalar device operator (no matrix export)"""
        nr, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_operator_shape(self._h, op, C.byref(nr), C.byref(ncol), C.byref(nnz)))
        out = np.empty(nr.value, dtype=np.float64)
        self._check(self._lib.nsfem_operator_diagonal(self._h, op, _dp(out)))
        return out

    def operator_nnz(self, op):
        """block-nonzeros x block size of a device operator"""
        nr, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_operator_shape(self._h, op, C.byref(nr), C.byref(ncol),
                                                   C.byref(nnz)))
        return nnz.value

    def operator_apply(self, op, x):
        if op == OP_MOMENTUM_JAC_MF:                 # matrix-free Jacobian at u = USTAR
            x = np.ascontiguousarray(x, dtype=np.float64)
            assert x.size == self.n_velocity
            y = np.empty(self.n_velocity, dtype=np.float64)
            self._check(self._lib.nsfem_operator_apply(self._h, op, _dp(x), _dp(y)))
            return y
        nr, ncol, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_operator_shape(self._h, op, C.byref(nr), C.byref(ncol),
                                                   C.byref(nnz)))
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == ncol.value
        y = np.empty(nr.value, dtype=np.float64)
        self._check(self._lib.nsfem_operator_apply(self._h, op, _dp(x), _dp(y)))
        return y

    def kernel_apply(self, space, nv, x, a=1.0, b_coef=0.0, family=0, epilogue=0, b=None, d=None, mask=None,
                     maskmode=0, steps=0, c1=(), c2=(), ghost=0, ident=False, from_zero=False,
                     with_residual=False, dict_ok=True):
        """test hook (nsfem_kernel_apply): product / residual / smoothing sequence of a M + b K through a chosen
        kernel family; returns dict(y, d, r, used_family, dict_entries, dict_exact, lattice_w)"""
        t = KernelTest()
        t.space, t.nv, t.family, t.epilogue, t.steps = int(space), int(nv), int(family), int(epilogue), int(steps)
        t.maskmode, t.ghost, t.ident = int(maskmode), int(ghost), 1 if ident else 0
        t.from_zero, t.with_residual, t.dict_ok = 1 if from_zero else 0, 1 if with_residual else 0, 1 if dict_ok else 0
        t.a, t.b_coef = float(a), float(b_coef)
        for k, v in enumerate(c1):
            t.c1[k] = float(v)
        for k, v in enumerate(c2):
            t.c2[k] = float(v)
        keep = []

        def arr(v, dtype=np.float64):
            if v is None:
                return None
            v = np.ascontiguousarray(v, dtype=dtype)
            keep.append(v)
            return v
        x = arr(x)
        n = x.size
        bb, dd, mm = arr(b), arr(d), arr(mask, np.uint8)
        y, d_out, r_out = np.empty(n), np.empty(n), np.empty(n)
        t.x, t.y, t.d_out, t.r_out = _dp(x), _dp(y), _dp(d_out), _dp(r_out)
        if bb is not None:
            assert bb.size == n
            t.b = _dp(bb)
        if dd is not None:
            assert dd.size == n
            t.d = _dp(dd)
        if mm is not None:
            assert mm.size == n
            t.mask = mm.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(self._lib.nsfem_kernel_apply(self._h, C.byref(t)))
        return dict(y=y, d=d_out, r=r_out, used_family=int(t.used_family), dict_entries=int(t.dict_entries),
                    dict_exact=bool(t.dict_exact), lattice_w=int(t.lattice_w))

    def set_preconditioner_shift(self, shift):
        self._check(self._lib.nsfem_set_preconditioner_shift(self._h, float(shift)))

    def set_angular_velocity(self, omega, omega_dot=0.0):
        if np.ndim(omega) > 0:                       # 3D: vectors
            w = np.ascontiguousarray(omega, dtype=np.float64)
            wd = np.ascontiguousarray(omega_dot if np.ndim(omega_dot) > 0 else np.zeros(3), dtype=np.float64)
            assert w.shape == (3, ) and wd.shape == (3, )
            self._check(self._lib.nsfem_set_angular_velocity_3d(self._h, _dp(w), _dp(wd)))
            return
        self._check(self._lib.nsfem_set_angular_velocity(self._h, float(omega), float(omega_dot)))

    def boundary_force(self, facet_cell, facet_local, nu, symmetric=1.0, velocity_slot=U0,
                       pressure_slot=P):
        """(force [dim], flux, measure) over the given boundary facets:
        force = int (-p n + nu (grad u + symmetric grad u^T) n) dS, flux = int u.n dS"""
        fc = np.ascontiguousarray(facet_cell, dtype=np.int32)
        fl = np.ascontiguousarray(facet_local, dtype=np.int32)
        assert fc.shape == fl.shape and fc.ndim == 1
        out = np.zeros(self.dim + 2)
        self._check(self._lib.nsfem_boundary_force(self._h, int(velocity_slot), int(pressure_slot),
                                                   fc.size, _ip(fc), _ip(fl), float(nu),
                                                   float(symmetric), _dp(out)))
        return out[:self.dim].copy(), float(out[self.dim]), float(out[self.dim + 1])

    def cfl_number(self, slot, step_size):
        out = C.c_double()
        self._check(self._lib.nsfem_cfl_number(self._h, int(slot), float(step_size), C.byref(out)))
        return out.value

    def profile_smoother(self, enable):
        """start (True) / stop (False -> (avg ms per launch, launches, algorithmic bytes per launch))
        the in-situ timing of the finest-level smoothing launches of the velocity multigrid"""
        ms, n, nbytes = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_profile_smoother(self._h, 1 if enable else 0, C.byref(ms),
                                                     C.byref(n), C.byref(nbytes)))
        return None if enable else (ms.value, n.value, nbytes.value)

    def profile_smoother_detail(self):
        """the window profile_smoother(False) just closed: dict(launches, steps, bytes, lattice)"""
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_profile_smoother_detail(self._h, out))
        return dict(launches=int(out[0]), steps=int(out[1]), bytes=int(out[2]), lattice=bool(out[3]))

    def jacobian_info(self):
        """the matrix-free velocity Jacobian action: dict(path = "three-launch" | "fused-gather" | "lattice-kernel",
        lattice_launches, lattice_bytes)"""
        out = (C.c_int64 * 4)()
        self._check(self._lib.nsfem_jacobian_info(self._h, out))
        return dict(path=("three-launch", "fused-gather", "lattice-kernel")[int(out[0])],
                    lattice_launches=int(out[1]), lattice_bytes=int(out[2]))

    def profile_convection(self, enable):
        """start (True) / stop (False -> (avg ms per application, applications, algorithmic bytes))
        the in-situ timing of the matrix-free convection action (element kernel + node gather)"""
        ms, n, nbytes = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self._lib.nsfem_profile_convection(self._h, 1 if enable else 0, C.byref(ms),
                                                       C.byref(n), C.byref(nbytes)))
        return None if enable else (ms.value, n.value, nbytes.value)

    def time_spmv(self, op, reps=50):
        ms = C.c_double()
        nbytes = C.c_int64()
        self._check(self._lib.nsfem_time_spmv(self._h, op, reps, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value
