"""ctypes wrapper around the CPU oracle (oracle/c/liborc.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by fesom2_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(REPO, "oracle", "c")
ORC_LIB = os.path.join(ORC_DIR, "liborc.so")
PD = C.POINTER(C.c_double)


def build():
    subprocess.run(["make", "-C", ORC_DIR, "-s"], check=True)
    return ORC_LIB


class Oracle:
    def __init__(self, mesh, params):
        if not os.path.exists(ORC_LIB):
            build()
        self.lib = C.CDLL(ORC_LIB)
        self.lib.orc_field_count.restype = C.c_longlong
        self.lib.orc_solver_residual.restype = C.c_double
        self.mesh, self.params = mesh, params
        rc = self.lib.orc_init(mesh.desc_p, C.byref(params))
        assert rc == 0

    def count(self, name):
        n = self.lib.orc_field_count(name.encode())
        if n < 0:
            raise KeyError(name)
        return n

    def set(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        rc = self.lib.orc_set_field(name.encode(), a.ctypes.data_as(PD), C.c_longlong(a.size))
        assert rc == 0, name

    def get(self, name, shape=None):
        n = self.count(name)
        out = np.empty(n, dtype=np.float64)
        rc = self.lib.orc_get_field(name.encode(), out.ctypes.data_as(PD), C.c_longlong(n))
        assert rc == 0, name
        return out.reshape(shape) if shape is not None else out

    def call(self, name, arg=0):
        rc = self.lib.orc_call(name.encode(), int(arg))
        assert rc == 0, name

    def step_info(self):
        """write_step_info + check_blowup (42 values, field order of fesom_step_info)"""
        out = np.empty(42, dtype=np.float64)
        assert self.lib.orc_step_info(out.ctypes.data_as(PD)) == 0
        return out

    def set_state(self, st):
        for k, v in st.a.items():
            self.set(k, v)
        if getattr(self.params, "use_density_ref", 0):          # ocean_setup: init_ref_density from the initial Z_3d_n (oce_setup_step.F90:129)
            self.call("init_ref_density")

    def first_step_done(self, v):
        self.lib.orc_set_first_step_done(int(v))

    @property
    def solver_iterations(self):
        return self.lib.orc_solver_iterations()

    @property
    def solver_residual(self):
        return self.lib.orc_solver_residual()
