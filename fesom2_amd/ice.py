"""Host-side handle on the sea-ice dynamics of libfesom_gpu.so (fesom_gpu_ice_*, include/fesom_gpu.h): the subcycled
momentum solve EVPdynamics_m of the reference (src/ice_maEVP.F90:273-602; whichEVP = 2: the adaptive EVPdynamics_a, :785-888; whichEVP = 0: the classic EVPdynamics, src/ice_EVP.F90:397-667) and the FCT advection of the ice fields (src/ice_fct.F90).
No CPU fallback."""
import ctypes as C
import numpy as np
from . import _lib


def ice_params(ice_dt=900.0, ellipse=2.0, alpha_evp=250.0, beta_evp=250.0, Pstar=30000.0, c_pressure=20.0, delta_min=1.0e-11, cd_oce_ice=5.5e-3,
               max_ice_loading=5.0, evp_rheol_steps=120, use_floatice=False, ice_gamma_fct=0.25, whichEVP=1, c_aevp=0.15, theta_io=0.0, Tevp_inv=None):
    """defaults = src/ice_modules.F90:7-27 (i_PARAM) and gen_modules_config.F90:67"""
    p = _lib.IceParams()
    p.ice_dt, p.ellipse, p.alpha_evp, p.beta_evp, p.Pstar, p.c_pressure = ice_dt, ellipse, alpha_evp, beta_evp, Pstar, c_pressure
    p.delta_min, p.cd_oce_ice, p.max_ice_loading = delta_min, cd_oce_ice, max_ice_loading
    p.evp_rheol_steps, p.use_floatice = int(evp_rheol_steps), int(use_floatice)
    p.ice_gamma_fct = ice_gamma_fct
    p.theta_io, p.Tevp_inv = theta_io, (3.0 / ice_dt if Tevp_inv is None else Tevp_inv)      # classic EVP (whichEVP = 0): src/ice_modules.F90:32, ice_setup_step.F90:33
    p.whichEVP, p.c_aevp = int(whichEVP), c_aevp          # 2 = adaptive EVP (EVPdynamics_a, src/ice_maEVP.F90:785-888; c_aevp: ice_modules.F90:36)
    return p


class IceFields:
    """the arrays of fesom_ice_state as contiguous float64 numpy arrays + the ctypes struct pointing at them"""

    def __init__(self, **arrays):
        self.a = {k: np.ascontiguousarray(v, dtype=np.float64).copy() for k, v in arrays.items()}
        self.desc = _lib.IceState()
        for k in _lib.ICE_FIELDS:
            setattr(self.desc, k, self.a[k].ctypes.data_as(_lib.PD) if k in self.a else None)

    def __getitem__(self, k):
        return self.a[k]


class IceCore:
    def __init__(self, mesh, params):
        self.lib = _lib.load()
        self.lib.fesom_gpu_ice_init.argtypes = [C.POINTER(_lib.MeshDesc), C.POINTER(_lib.PartDesc), C.POINTER(_lib.IceParams)]
        self.lib.fesom_gpu_ice_upload.argtypes = [C.POINTER(_lib.IceState)]
        self.lib.fesom_gpu_ice_download.argtypes = [C.POINTER(_lib.IceState)]
        self.lib.fesom_gpu_ice_evp.argtypes = [C.c_int]
        self.lib.fesom_gpu_ice_time_ms.argtypes = [C.c_int, _lib.PD]
        self.lib.fesom_gpu_ice_last_error.restype = C.c_char_p
        self.mesh, self.params = mesh, params
        self._chk(self.lib.fesom_gpu_ice_init(mesh.desc_p, mesh.part_p, C.byref(params)), "ice_init")

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.fesom_gpu_ice_last_error().decode()}")

    def upload(self, fields):
        self._chk(self.lib.fesom_gpu_ice_upload(C.byref(fields.desc)), "ice_upload")

    def evp(self, ncalls=1):
        self._chk(self.lib.fesom_gpu_ice_evp(int(ncalls)), "ice_evp")

    def advect(self, ncalls=1):
        """FCT advection of m_ice, a_ice, m_snow with the current ice velocities (src/ice_fct.F90 + cut_off)"""
        self.lib.fesom_gpu_ice_advect.argtypes = [C.c_int]
        self._chk(self.lib.fesom_gpu_ice_advect(int(ncalls)), "ice_advect")

    def step(self, nsteps=1):
        """the dynamics of ice_timestep: EVPdynamics_m, then the advection part (src/ice_setup_step.F90:195-232)"""
        for _ in range(int(nsteps)):
            self.evp(1); self.advect(1)

    def download(self, fields):
        self._chk(self.lib.fesom_gpu_ice_download(C.byref(fields.desc)), "ice_download")

    def time_ms(self, ncalls=5):
        ms = C.c_double(0.0)
        self._chk(self.lib.fesom_gpu_ice_time_ms(int(ncalls), C.byref(ms)), "ice_time_ms")
        return ms.value

    def close(self):
        self.lib.fesom_gpu_ice_finalize()
