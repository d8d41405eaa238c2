"""Host-side handle on the MI355X ocean core: thin Python mirror of the C ABI
(fesom_gpu_init / upload_state / step / download_state ...).  The call surface follows the
reference's time loop: `step(n)` = compute_vel_nodes + oce_timestep_ale(n, mesh)
(src/fvom_main.F90:216,250).  No CPU fallback exists: every method goes through libfesom_gpu.so."""
import ctypes as C
import numpy as np
from . import _lib


class OceanCore:
    def __init__(self, mesh, params):
        self.lib = _lib.load()
        self.mesh, self.params = mesh, params
        rc = self.lib.fesom_gpu_init(mesh.desc_p, mesh.part_p, C.byref(params))
        if rc != 0:
            raise RuntimeError(f"fesom_gpu_init failed ({rc}): {self.lib.fesom_gpu_last_error().decode()}")

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.fesom_gpu_last_error().decode()}")

    def upload_state(self, state):
        d = state.desc()
        self._chk(self.lib.fesom_gpu_upload_state(C.byref(d)), "upload_state")

    def download_state(self, state):
        d = state.desc()
        self._chk(self.lib.fesom_gpu_download_state(C.byref(d)), "download_state")

    def set_forcing(self, **arrays):
        f = _lib.ForcingDesc()
        keep = []
        for k, v in arrays.items():
            a = np.ascontiguousarray(v, dtype=np.float64)
            keep.append(a)
            setattr(f, k, a.ctypes.data_as(_lib.PD))
        self._chk(self.lib.fesom_gpu_set_forcing(C.byref(f)), "set_forcing")

    def call(self, routine, arg=0):
        self._chk(self.lib.fesom_gpu_call(routine.encode(), int(arg)), f"call({routine})")

    def get(self, name, count, shape=None):
        out = np.empty(int(count), dtype=np.float64)
        self._chk(self.lib.fesom_gpu_get_field(name.encode(), out.ctypes.data_as(_lib.PD), C.c_longlong(out.size)), f"get({name})")
        return out.reshape(shape) if shape is not None else out

    def set(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.lib.fesom_gpu_set_field(name.encode(), a.ctypes.data_as(_lib.PD), C.c_longlong(a.size)), f"set({name})")

    def step(self, n=1):
        self._chk(self.lib.fesom_gpu_step(int(n)), "step")

    def run_steps(self, n_first, nsteps):
        """Enqueue nsteps steps back to back (no host synchronisation)."""
        self._chk(self.lib.fesom_gpu_run_steps(int(n_first), int(nsteps)), "run_steps")

    def sync(self):
        """wait for the enqueued steps; deferred device-side errors (zlevel: the missing local-zstar fallback) are reported here"""
        self._chk(self.lib.fesom_gpu_sync(), "sync")

    def step_info(self):
        """device-side step monitor (write_step_info + check_blowup of the reference) over the owned nodes; returns a dict"""
        si = _lib.StepInfo()
        self._chk(self.lib.fesom_gpu_step_info(C.byref(si)), "step_info")
        return {n: getattr(si, n) for n in _lib.STEP_INFO_FIELDS}

    def profile_step(self, n=1):
        """one step, phase by phase; returns the device ms of the reference's phase timers (mixpres, dyn, dynssh, solvessh, GMRedi, solvetra, total)"""
        ms = (C.c_double * 7)()
        self._chk(self.lib.fesom_gpu_profile_step(int(n), ms), "profile_step")
        return dict(zip(("mixpres", "dyn", "dynssh", "solvessh", "GMRedi", "solvetra", "total"), list(ms)))

    def kernel_time_ms(self, group, nrep=20):
        ms = C.c_double(0.0)
        self._chk(self.lib.fesom_gpu_kernel_time_ms(group.encode(), int(nrep), C.byref(ms)), f"kernel_time_ms({group})")
        return ms.value

    @property
    def solver_iterations(self):
        return self.lib.fesom_gpu_last_solver_iterations()

    @property
    def tile_shape(self):
        return self.lib.fesom_gpu_tile_shape()

    @property
    def solver_residual(self):
        return self.lib.fesom_gpu_last_solver_residual()

    def close(self):
        self.lib.fesom_gpu_finalize()
