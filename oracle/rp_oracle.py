"""ctypes wrapper for the CPU oracle (TEST INFRASTRUCTURE — see rp_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  PARITY UNPINNED (no MuJoCo in the reference tree or image).
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librp_oracle.so")

FIELDS = dict(
    qpos=0, qvel=1, qacc=2, qacc_warmstart=3, ctrl=4, qfrc_applied=5,
    actuator_force=6, actuator_velocity=7, actuator_length=8, xpos=9, xmat=10,
    geom_xpos=11, geom_xmat=12, site_xpos=13, qM=14, qfrc_bias=15, qfrc_passive=16,
    qfrc_actuator=17, qfrc_smooth=18, qacc_smooth=19, qfrc_constraint=20,
    efc_force=21, efc_aref=22, efc_D=23, efc_pos=24, efc_J=25, contact=26, time=27,
    body_pos=28, sensor_torque=29, sensor_touch=30, cfrc_int=31, subtree_com=32,
)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rp_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librp_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.rpo_model_load.restype = ctypes.c_void_p
        L.rpo_model_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.rpo_model_free.argtypes = [ctypes.c_void_p]
        L.rpo_data_new.restype = ctypes.c_void_p
        L.rpo_data_new.argtypes = [ctypes.c_void_p]
        L.rpo_data_free.argtypes = [ctypes.c_void_p]
        for f in ("rpo_reset", "rpo_forward", "rpo_step", "rpo_step1"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.rpo_get_ptr.restype = ctypes.POINTER(ctypes.c_double)
        L.rpo_get_ptr.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for f in ("rpo_ncon", "rpo_nefc", "rpo_solver_iter", "rpo_warnings"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = ctypes.c_int
        L.rpo_bench.restype = ctypes.c_double
        L.rpo_bench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.rpo_bench_seq.restype = ctypes.c_double
        L.rpo_bench_seq.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.rpo_debug_set_mpr.argtypes = [ctypes.c_double, ctypes.c_int]
        L.rpo_debug_set_mpr_poly.argtypes = [ctypes.c_double]
        L.rpo_debug_set_capsule_box.argtypes = [ctypes.c_int]
        L.rpo_debug_set_boxbox_max.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


MPR_POLY_REFINED = 1e-10   # the polytope-pair tolerance of the step-by-step comparisons (engine: rp_set_mpr_tolerance)


def set_mpr_experiment(tolerance: float = 1e-6, discrete: bool = False, poly_tolerance: float | None = None) -> None:
    """Stopping rule of the hull / cylinder narrow phase (process-wide).  No arguments = MuJoCo's uniform rule (1e-6 for
    every pair), which is the oracle's default.  `poly_tolerance`: polytope pairs (box / hull on both sides) refine to
    that instead -- converged, two implementations cannot stop on different portals (rp_oracle.c: g_mpr_tol_poly);
    `discrete` = polytope pairs stop when the support vertex already is a portal vertex (a tolerance-free rule)."""
    lib().rpo_debug_set_mpr(float(tolerance), int(bool(discrete)))
    lib().rpo_debug_set_mpr_poly(-1.0 if poly_tolerance is None else float(poly_tolerance))


# The narrow-phase choices this restatement could not pin from memory, as process-wide switches, so
# that the first real-MuJoCo recording that disagrees with the defaults can be bisected (oracle/bisect_golden.py).
# Defaults = what the HIP engine computes; any other value is an ORACLE-ONLY experiment.
NARROW_PHASE_VARIANTS = {
    "capsule_box": {0: "closest axis point + the deeper end (default)", 1: "closest axis point only",
                    2: "the two ends only", 3: "closest axis point + both ends"},
    "boxbox_max": {8: "all clipped points, up to eight (default)", 4: "the first four", 3: "the first three", 1: "one point"},
    "mpr": {"uniform": "1e-6 for every pair (MuJoCo's rule; default)", "refined": "1e-6, polytope pairs at 1e-10",
            "discrete": "polytope pairs stop at a repeated support vertex", "tight": "refinement stops at 1e-10"},
}


def set_narrow_phase_variant(capsule_box: int = 0, boxbox_max: int = 8, mpr: str = "uniform") -> None:
    """Selects one combination of NARROW_PHASE_VARIANTS (no arguments = the defaults)."""
    if capsule_box not in NARROW_PHASE_VARIANTS["capsule_box"] or mpr not in NARROW_PHASE_VARIANTS["mpr"] or not 1 <= boxbox_max <= 8:
        raise ValueError("unknown narrow-phase variant")
    L = lib()
    L.rpo_debug_set_capsule_box(int(capsule_box))
    L.rpo_debug_set_boxbox_max(int(boxbox_max))
    set_mpr_experiment(1e-10 if mpr == "tight" else 1e-6, mpr == "discrete", MPR_POLY_REFINED if mpr == "refined" else None)


def first_crossing(rel, level=1e-6):
    """First mj_step (1-based) at which the per-step relative error curve `rel` exceeds `level`; 0 = never."""
    idx = np.nonzero(np.asarray(rel) > level)[0]
    return int(idx[0]) + 1 if len(idx) else 0


def chaos_control(model, blob, ctrl_seq, nstep=1000, hold=10, seeds=(0, 1, 2), eps0=1e-15, eps_step=0.0,
                  marks=(1, 10, 100, 300, 1000), cross_level=1e-6):
    """The CONTROL of the free-running parity figure: the oracle against ITSELF with a rounding-sized perturbation,
    on the action stream `ctrl_seq` [T, nu] (one row per `hold` mj_steps) from the reset state.
      eps0     one-time perturbation of qpos after the reset: qpos += eps0 * N(0, 1)
      eps_step per-step multiplicative noise on qvel: qvel *= 1 + eps_step * N(0, 1) (the size of a second
               implementation's per-step rounding difference; the engine's measured teacher-forced discrepancy is
               ~3e-12 of the step's velocity change)
    Returns, per seed, the same figures bench.py's parity block prints for engine-vs-oracle:
    rel = |dq| / max(|q_ref|, 1e-2), its maximum over the run and the running maximum at `marks`."""
    base = Oracle(model, blob)
    base.reset()
    ref = np.zeros((nstep, model.nv))
    for i in range(nstep):
        base.ctrl[:] = ctrl_seq[(i // hold) % ctrl_seq.shape[0]]
        base.step(1)
        ref[i] = base.qpos
    out = []
    for seed in seeds:
        rng = np.random.default_rng(seed)
        o = Oracle(model, blob)
        o.reset()
        o.qpos[:] += eps0 * rng.standard_normal(model.nv)
        worst, curve, cross = 0.0, {}, 0
        for i in range(nstep):
            o.ctrl[:] = ctrl_seq[(i // hold) % ctrl_seq.shape[0]]
            o.step(1)
            if eps_step:
                o.qvel[:] *= 1.0 + eps_step * rng.standard_normal(model.nv)
            worst = max(worst, float((np.abs(o.qpos - ref[i]) / np.maximum(np.abs(ref[i]), 1e-2)).max()))
            if not cross and worst > cross_level:
                cross = i + 1   # (the mj_step at which this control leaves the reference trajectory: the chaotic event)
            if i + 1 in marks:
                curve[str(i + 1)] = worst
        out.append({"seed": int(seed), "max_rel_qpos_error": worst, "running_max_at_mj_step": curve,
                    "first_mj_step_above_%g" % cross_level: cross})
    return out


class Oracle:
    """One fp64 environment stepped by the CPU oracle."""

    def __init__(self, model, blob: bytes):
        self.m = model
        self._L = lib()
        self._blob = blob
        self._model = self._L.rpo_model_load(blob, len(blob))
        if not self._model:
            raise RuntimeError("rpo_model_load failed (bad blob)")
        self._data = self._L.rpo_data_new(self._model)
        self.reset()

    def __del__(self):
        try:
            self._L.rpo_data_free(self._data)
            self._L.rpo_model_free(self._model)
        except Exception:
            pass

    def _size(self, name):
        m = self.m
        return dict(
            qpos=m.nv, qvel=m.nv, qacc=m.nv, qacc_warmstart=m.nv, ctrl=m.nu,
            qfrc_applied=m.nv, actuator_force=m.nu, actuator_velocity=m.nu,
            actuator_length=m.nu, xpos=3 * m.nbody, xmat=9 * m.nbody,
            geom_xpos=3 * m.ngeom, geom_xmat=9 * m.ngeom, site_xpos=3 * m.nsite,
            qM=m.nv * m.nv, qfrc_bias=m.nv, qfrc_passive=m.nv, qfrc_actuator=m.nv,
            qfrc_smooth=m.nv, qacc_smooth=m.nv, qfrc_constraint=m.nv,
            efc_force=self.nefc, efc_aref=self.nefc, efc_D=self.nefc,
            efc_pos=self.nefc, efc_J=self.nefc * m.nv, contact=16 * self.ncon, time=1,
            body_pos=3 * m.nbody, sensor_torque=m.nv, sensor_touch=m.nsite, cfrc_int=6 * m.nbody,
            subtree_com=3 * m.nbody,
        )[name]

    def view(self, name) -> np.ndarray:
        """Writable numpy view of an oracle array."""
        n = self._size(name)
        p = self._L.rpo_get_ptr(self._model, self._data, FIELDS[name])
        if n == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(p, shape=(n,))

    def __getattr__(self, name):
        if name in FIELDS:
            return self.view(name)
        raise AttributeError(name)

    @property
    def ncon(self):
        return self._L.rpo_ncon(self._data)

    @property
    def nefc(self):
        return self._L.rpo_nefc(self._data)

    @property
    def solver_iter(self):
        return self._L.rpo_solver_iter(self._data)

    @property
    def warnings(self):
        return self._L.rpo_warnings(self._data)

    def reset(self):
        self._L.rpo_reset(self._model, self._data)

    def forward(self):
        self._L.rpo_forward(self._model, self._data)

    def step1(self):
        """mj_step1 of the CURRENT qpos / qvel (position / velocity stage only; qacc_warmstart untouched): call it after
        writing a state from outside -- `step` runs mj_step2 on the stage data in place, i.e. on the data of whatever
        state the previous call left."""
        self._L.rpo_step1(self._model, self._data)

    def step(self, n: int = 1):
        for _ in range(n):
            self._L.rpo_step(self._model, self._data)

    def bench_seq(self, nenv, nstep, ctrl_seq, start, hold=10, nthreads=0):
        """Every env replays `ctrl_seq` [T, nu] from its own row start[e], one row per `hold` mj_steps.
        Returns (seconds, qpos[nenv, nv])."""
        out = np.zeros((nenv, self.m.nv))
        c = np.ascontiguousarray(ctrl_seq, np.float64)
        st = np.ascontiguousarray(start, np.int32)
        assert c.shape[1] == self.m.nu and st.shape == (nenv,)
        t = self._L.rpo_bench_seq(self._model, nenv, nstep, c.ctypes.data, c.shape[0], hold, st.ctypes.data,
                                  nthreads, out.ctypes.data)
        return t, out

    def bench(self, nenv, nstep, ctrl=None, nthreads=0):
        """Returns (seconds, qpos[nenv, nv])."""
        out = np.zeros((nenv, self.m.nv))
        c = None
        if ctrl is not None:
            c = np.ascontiguousarray(ctrl, np.float64)
            assert c.shape == (nenv, self.m.nu)
        t = self._L.rpo_bench(
            self._model, nenv, nstep,
            c.ctypes.data if c is not None else None, nthreads, out.ctypes.data)
        return t, out
