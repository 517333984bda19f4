"""ctypes binding of the HIP engine's C ABI (include/rp_engine.h).

`BatchedPhysics` plays the role `dm_control.mjcf.Physics` plays for the
reference (call sites listed in SURVEY.md §8b): named, batched views of
qpos/qvel/ctrl/sensors/site positions plus `step()`/`forward()`/`reset()`.
There is NO CPU fallback: if the shared library is missing or no HIP device
is present, construction raises.
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

from robopianist_amd.model import compile as mcompile
from robopianist_amd.model import engine_tables

_HERE = os.path.dirname(os.path.abspath(__file__))
# RP_ENGINE_LIB selects another build of the same library (perf experiments: scratch/run_variants.py)
LIB_PATH = os.environ.get("RP_ENGINE_LIB") or os.path.join(_HERE, "csrc", "librp_engine.so")

# rp_field
QPOS, QVEL, QACC_WARMSTART, CTRL, QFRC_APPLIED, ACT_FORCE, ACT_VELOCITY, SITE_XPOS, \
    TIME, NCON, CONTACT_GEOMS, WARN_FLAGS, SOLVER_ITER, CONTACT_DIST, TREE_OFFSET, ACTIVE, \
    SENSOR_TORQUE, SENSOR_TOUCH, ENV_COST, DEBUG_MASS_ROWS, DEBUG_HANDOVER_HDR = range(21)
MAX_CONTACTS = 64

WARN_BADSTATE = 1
WARN_CONTACT_FULL = 2
WARN_HESSIAN = 4
WARN_KEYSLOT_FULL = 8
WARN_WORK_FULL = 16
WARN_SPLIT_FULL = 64
WARN_DENSE_FULL = 32

EXPORTED_SYMBOLS = (
    "rp_create", "rp_destroy", "rp_reset", "rp_set", "rp_get", "rp_step", "rp_forward", "rp_step_masked",
    "rp_set_solver_limits", "rp_set_solver_tolerance", "rp_set_mpr_tolerance", "rp_set_lazy_position_stage", "rp_set_legacy_step", "rp_set_cost_ordered_launch", "rp_set_stream_slices", "rp_set_lean_solver", "rp_set_fused_substeps", "rp_get_fused_substeps", "rp_set_split_position_stage", "rp_get_split_position_stage", "rp_set_acc_sensors", "rp_sync", "rp_get_stream", "rp_set_stream", "rp_field_ptr",
    "rp_n_envs", "rp_dim",
    "rp_kernel_time", "rp_solver_kernel_time", "rp_solver_kernel_envs", "rp_solver_kernel_fused", "rp_profile", "rp_last_error",
)

_lib = None


class EngineError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """Loads librp_engine.so; raises EngineError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise EngineError(
            f"HIP engine library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    # torch bundles its own libamdhip64; it has to be in the process BEFORE this library pulls in the
    # system one (same soname: the loader then binds us to torch's copy and the two share one runtime,
    # which is what makes rp_field_ptr views and stream sharing work).  The other order leaves torch
    # with "No HIP GPUs are available" at its first CUDA call.
    try:
        import torch  # noqa: F401
    except ImportError:  # plain ctypes use without torch is fine
        pass
    L = ctypes.CDLL(path)
    L.rp_last_error.restype = ctypes.c_char_p
    L.rp_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                            ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.rp_destroy.argtypes = [ctypes.c_void_p]
    L.rp_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.rp_set.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.rp_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.rp_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.rp_forward.argtypes = [ctypes.c_void_p]
    L.rp_step_masked.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.rp_set_solver_limits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.rp_sync.argtypes = [ctypes.c_void_p]
    L.rp_set_solver_tolerance.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
    L.rp_set_mpr_tolerance.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
    L.rp_set_lazy_position_stage.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_cost_ordered_launch.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_acc_sensors.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_stream_slices.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_lean_solver.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_legacy_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_set_fused_substeps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_get_fused_substeps.argtypes = [ctypes.c_void_p]
    L.rp_set_split_position_stage.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.rp_get_split_position_stage.argtypes = [ctypes.c_void_p]
    L.rp_get_stream.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    L.rp_n_envs.argtypes = [ctypes.c_void_p]
    L.rp_dim.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.rp_kernel_time.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_int)]
    L.rp_solver_kernel_time.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int)]
    L.rp_solver_kernel_envs.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    L.rp_solver_kernel_fused.argtypes = [ctypes.c_void_p]
    L.rp_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.rp_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.rp_field_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_size_t)]
    _lib = L
    return L


def make_blob(model: mcompile.Model, key_joint_ids: np.ndarray) -> bytes:
    """Model blob including the engine tables (what rp_create expects)."""
    tables = engine_tables.build_engine_tables(model, key_joint_ids)
    return mcompile.to_blob(model, extra=tables)


def _ptr(x):
    """Raw address of a numpy array or a torch tensor (host or device)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return x.data_ptr()
    raise TypeError(type(x))


class BatchedPhysics:
    """`n_envs` copies of one compiled scene stepped on one MI355X."""

    def __init__(self, model: mcompile.Model, key_joint_ids: np.ndarray, n_envs: int,
                 device_id: int = 0, precision: int = 64, blob: Optional[bytes] = None,
                 self_check: Optional[bool] = None):
        self._L = load_library()
        self.model = model
        self.n_envs = int(n_envs)
        self.device_id = int(device_id)
        self.precision = int(precision)
        self.dtype = np.float32 if precision == 32 else np.float64
        self.blob = blob if blob is not None else make_blob(model, key_joint_ids)
        self._h = ctypes.c_void_p()
        rc = self._L.rp_create(self.blob, len(self.blob), self.n_envs, int(device_id),
                               self.precision, ctypes.byref(self._h))
        if rc != 0:
            raise EngineError(self._L.rp_last_error().decode())
        self.nv = self.dim("nv"); self.nu = self.dim("nu"); self.nsite = self.dim("nsite")
        self.ntree = self.dim("ntree")
        self._shapes = {
            QPOS: (self.nv,), QVEL: (self.nv,), QACC_WARMSTART: (self.nv,), CTRL: (self.nu,),
            QFRC_APPLIED: (self.nv,), ACT_FORCE: (self.nu,), ACT_VELOCITY: (self.nu,),
            SITE_XPOS: (self.nsite, 3), TIME: (), NCON: (), CONTACT_GEOMS: (MAX_CONTACTS, 2),
            WARN_FLAGS: (), SOLVER_ITER: (), CONTACT_DIST: (MAX_CONTACTS,),
            TREE_OFFSET: (self.ntree, 3), ACTIVE: (), SENSOR_TORQUE: (self.nv,), SENSOR_TOUCH: (self.nsite,),
            ENV_COST: (), DEBUG_HANDOVER_HDR: (8,),
        }
        self._shapes[DEBUG_MASS_ROWS] = (self.dim("rm_rows"), self.dim("rm_cols"))   # (the kernel build in use decides)
        self._int_fields = {NCON, CONTACT_GEOMS, WARN_FLAGS, SOLVER_ITER, ACTIVE, ENV_COST, DEBUG_HANDOVER_HDR}
        if self_check is None:
            self_check = os.environ.get("RP_SKIP_SELF_CHECK", "0") != "1"
        if self_check:
            _build_self_check(model, key_joint_ids, self.blob, self.device_id)

    def __del__(self):
        try:
            if self._h:
                self._L.rp_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(self._L.rp_last_error().decode())

    def dim(self, name: str) -> int:
        return self._L.rp_dim(self._h, name.encode())

    def field_shape(self, f):
        return (self.n_envs,) + self._shapes[f]

    def field_dtype(self, f):
        return np.int32 if f in self._int_fields else self.dtype

    def reset(self, mask=None):
        """mask: None, a numpy uint8/bool array, or a torch uint8 tensor (device masks keep
        the call asynchronous)."""
        if mask is not None and isinstance(mask, np.ndarray):
            mask = np.ascontiguousarray(mask, np.uint8)
            assert mask.shape == (self.n_envs,)
        self._check(self._L.rp_reset(self._h, _ptr(mask)))

    def set_stream(self, hip_stream: int):
        self._check(self._L.rp_set_stream(self._h, ctypes.c_void_p(hip_stream)))

    def view(self, f):
        """Zero-copy torch tensor aliasing the engine's array for field `f`.  Reads / writes through the view run
        on torch's current stream: either make the engine step on that stream (`set_stream`, what
        `suite.physics.TorchPhysics` does) or order them yourself (`sync()` / `torch.cuda.synchronize()`)."""
        import torch
        p = ctypes.c_void_p(); nb = ctypes.c_size_t()
        self._check(self._L.rp_field_ptr(self._h, f, ctypes.byref(p), ctypes.byref(nb)))
        shape = self.field_shape(f)
        dt = self.field_dtype(f)

        class _Mem:
            pass
        mem = _Mem()
        mem.__cuda_array_interface__ = {
            "shape": shape, "typestr": np.dtype(dt).str, "data": (p.value or 0, False),
            "version": 2, "strides": None,
        }
        if int(np.prod(shape)) == 0:
            return torch.zeros(shape, dtype=getattr(torch, np.dtype(dt).name), device="cuda")
        t = torch.as_tensor(mem, device=torch.device("cuda", self.device_id))
        t._rp_owner = self  # keep the engine alive as long as the view
        return t

    def set(self, f, value):
        """value: numpy array or torch tensor (host or device) of the field's shape.
        set(ACTIVE, None) re-enables every env."""
        if value is None:
            self._check(self._L.rp_set(self._h, f, None))
            return
        if isinstance(value, np.ndarray) or not hasattr(value, "data_ptr"):
            value = np.ascontiguousarray(np.broadcast_to(
                np.asarray(value, self.field_dtype(f)), self.field_shape(f)))
        else:
            assert tuple(value.shape) == self.field_shape(f), (value.shape, self.field_shape(f))
        self._check(self._L.rp_set(self._h, f, _ptr(value)))
        # host sources must stay alive until the async copy has been consumed
        self._check(self._L.rp_sync(self._h)) if isinstance(value, np.ndarray) else None

    def get(self, f, out=None):
        if out is None:
            out = np.empty(self.field_shape(f), self.field_dtype(f))
        self._check(self._L.rp_get(self._h, f, _ptr(out)))
        return out

    def step(self, n_substeps: int = 1, key_trace=None):
        """key_trace: optional uint32 array/tensor [n_envs, n_substeps, 4]."""
        self._check(self._L.rp_step(self._h, int(n_substeps), _ptr(key_trace)))

    def forward(self):
        self._check(self._L.rp_forward(self._h))

    def step_masked(self, n_substeps: int, key_trace, reset_mask):
        """rp_step_masked: reset + forward of the envs `reset_mask` (uint8 device tensor [n_envs]) flags, then the
        step of the envs the RP_ACTIVE mask selects (include/rp_engine.h)."""
        self._check(self._L.rp_step_masked(self._h, int(n_substeps), _ptr(key_trace), _ptr(reset_mask)))

    def sync(self):
        self._check(self._L.rp_sync(self._h))

    def set_solver_limits(self, max_newton_iter=0, max_ls_iter=0):
        self._check(self._L.rp_set_solver_limits(self._h, max_newton_iter, max_ls_iter))

    def set_lazy_position_stage(self, on: bool = True):
        """rp_step skips its leading position/velocity stage for envs whose stage data is still
        valid (see include/rp_engine.h); state written through views then needs forward()."""
        self._check(self._L.rp_set_lazy_position_stage(self._h, int(bool(on))))

    def set_acc_sensors(self, on: bool = True):
        """Torque / touch sensors (SENSOR_TORQUE, SENSOR_TOUCH) after every step(): one extra launch per
        rp_step (include/rp_engine.h)."""
        self._check(self._L.rp_set_acc_sensors(self._h, int(bool(on))))

    def set_stream_slices(self, n: int = 0):
        """rp_step steps the batch as `n` (1 or 2; 0 = the engine picks) kernel chains (include/rp_engine.h)."""
        self._check(self._L.rp_set_stream_slices(self._h, int(n)))

    def set_legacy_step(self, on: bool = True):
        """dm_control's legacy_step (include/rp_engine.h: rp_set_legacy_step): False = the position-dependent outputs
        after step() are those of the state before the last integration (mj_step order)."""
        self._check(self._L.rp_set_legacy_step(self._h, int(bool(on))))

    def set_lean_solver(self, on=True):
        """Capacity classes of the solver stage (include/rp_engine.h: rp_set_lean_solver); an int > 1 caps the
        light class at that many contact Jacobian entries (tests: both classes in one small scene)."""
        self._check(self._L.rp_set_lean_solver(self._h, int(on)))

    def set_fused_substeps(self, on=True):
        """All substeps of a step in one launch (include/rp_engine.h: rp_set_fused_substeps): False / True, or
        "auto" (2): a candidate of the engine's own schedule choice."""
        self._check(self._L.rp_set_fused_substeps(self._h, 2 if on == "auto" else int(on)))

    @property
    def fused_substeps(self) -> bool:
        """True if rp_step currently runs the fused schedule."""
        return self._L.rp_get_fused_substeps(self._h) == 1

    def set_split_position_stage(self, on=True):
        """Position stage of the substeps as front part / pooled narrow phase / back part (include/rp_engine.h:
        rp_set_split_position_stage): False / True, or "auto" (2): a candidate of the engine's own schedule choice.
        Bit-identical results."""
        self._check(self._L.rp_set_split_position_stage(self._h, 2 if on == "auto" else int(on)))

    @property
    def split_position_stage(self) -> bool:
        """True if rp_step currently runs the split position stage."""
        return self._L.rp_get_split_position_stage(self._h) == 1

    def set_cost_ordered_launch(self, on: bool = True):
        """Stage kernels process the envs heaviest-first (include/rp_engine.h); bit-identical results."""
        self._check(self._L.rp_set_cost_ordered_launch(self._h, int(bool(on))))

    def set_mpr_tolerance(self, tolerance=0.0, polytope_tolerance=0.0):
        """Stopping tolerance of the hull / cylinder narrow phase (default: MuJoCo's uniform 1e-6);
        `polytope_tolerance` = box / hull pairs refine to that instead (include/rp_engine.h)."""
        self._check(self._L.rp_set_mpr_tolerance(self._h, float(tolerance), float(polytope_tolerance)))

    def set_solver_tolerance(self, tolerance=0.0, ls_tolerance=0.0):
        self._check(self._L.rp_set_solver_tolerance(self._h, float(tolerance), float(ls_tolerance)))

    def kernel_time(self):
        """(average device ms of one rp_step launch sequence since last call, sequences timed)."""
        ms = ctypes.c_double(); n = ctypes.c_int()
        self._check(self._L.rp_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def solver_kernel_time(self):
        """(average device ms of ONE solver-kernel launch since last call, launches sampled)."""
        ms = ctypes.c_double(); n = ctypes.c_int()
        self._check(self._L.rp_solver_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def solver_kernel_envs(self):
        """Average envs per launch over the launches the last solver_kernel_time() reported."""
        v = ctypes.c_double()
        self._check(self._L.rp_solver_kernel_envs(self._h, ctypes.byref(v)))
        return v.value

    def solver_kernel_fused(self) -> bool:
        """True when the launches the last solver_kernel_time() reported were fused-substeps launches."""
        return self._L.rp_solver_kernel_fused(self._h) == 1

    def profile(self, enable=True):
        """Reads+clears the per-phase cycle counters of env 0, then (dis)enables them."""
        out = np.zeros(64, np.int64)   # (RPK_NPROF_ALL: 0 .. 47 per-phase cycles of env 0, 48 .. 63 the pooled narrow phase's lists)
        self._check(self._L.rp_profile(self._h, out.ctypes.data, 64, int(bool(enable))))
        return out

    @property
    def stream(self) -> int:
        s = ctypes.c_void_p()
        self._check(self._L.rp_get_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    # convenience accessors (host copies)
    @property
    def qpos(self): return self.get(QPOS)
    @property
    def qvel(self): return self.get(QVEL)
    @property
    def ctrl(self): return self.get(CTRL)
    @property
    def time(self): return self.get(TIME)
    @property
    def warn_flags(self): return self.get(WARN_FLAGS)


_self_checked = set()


def _build_self_check(model, key_joint_ids, blob, device_id):
    """Loud failure instead of silently wrong physics if librp_engine.so was miscompiled.

    The fp64 solver kernel needs ~500 registers per lane; at that pressure hipcc has been
    seen (round 1, ROCm 7.2) to emit a wrong kernel after a harmless source edit (correct
    again with `-mllvm -amdgpu-spill-sgpr-to-vgpr=false`, at 2.6x the run time).  The fp32
    and fp64 builds are separate instantiations, so a few substeps of both on the same input
    must agree to single precision (measured: 2e-7 in q, 1e-5 in v; the two bad builds seen so far
    were off by 5e-4 and 1e-2); once per (library, model, device) and process."""
    key = (LIB_PATH, hash(blob), device_id)
    if key in _self_checked:
        return
    lo, hi = model.actuator_ctrlrange[:, 0], model.actuator_ctrlrange[:, 1]
    ctrl = (lo + 0.6 * (hi - lo))[None, :] if model.nu else np.zeros((1, 0))
    out = {}
    for prec in (64, 32):
        e = BatchedPhysics(model, key_joint_ids, 2, device_id=device_id, precision=prec, blob=blob,
                           self_check=False)
        if model.nu:
            e.set(CTRL, ctrl)
        e.step(10)
        out[prec] = (e.qpos.astype(np.float64)[0], e.get(QVEL).astype(np.float64)[0])
        del e
    dq = np.abs(out[64][0] - out[32][0]).max()
    dv = np.abs(out[64][1] - out[32][1]).max()
    vs = max(1.0, np.abs(out[64][1]).max())
    if os.environ.get("RP_SELF_CHECK_VERBOSE"):
        print(f"rp self-check: fp32 vs fp64 after 10 substeps: max|dq| = {dq:.2e}, max|dv| = {dv:.2e}")
    if not (np.isfinite(dq) and np.isfinite(dv) and dq < 2e-5 and dv < 2e-3 * vs):
        raise EngineError(
            f"librp_engine.so failed its build self-check: the fp32 and fp64 kernels disagree after 10 "
            f"substeps (max |dq| = {dq:.2e}, max |dv| = {dv:.2e}).  The library was most likely "
            "miscompiled; rebuild it, and see DESIGN.md (toolchain note).")
    _self_checked.add(key)


def decode_key_trace(trace: np.ndarray, n_keys: int = 88) -> np.ndarray:
    """[E, n_sub, 4] uint32 bit masks -> [E, n_sub, n_keys] bool."""
    t = np.asarray(trace, np.uint32)
    bits = (t[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1
    return bits.reshape(t.shape[:-1] + (128,))[..., :n_keys].astype(bool)
