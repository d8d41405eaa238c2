"""Named workloads of the hot path (mesh + options + initial state), shared by bench.py, the tools and the tests.

  pi       : BASELINE config #2 -- the reference's pi mesh (3140 nodes, 47 layers), analytic T/S, physics "default"
             (KPP + GM + Redi + analytic surface forcing = config/namelist.oce) or "pp".
  channel  : BASELINE config #3 stand-in (the reference ships no CORE2 mesh) -- the Soufflet channel of the reference's
             CI case test_souf refined `levels` times (3 -> 184 000 nodes, 2.5 km), 47 stretched layers, the toy's own
             options (linear EOS, PP mixing, zonal relaxation hooks), dt = 1200 s / 2**levels.  See channel_mesh.py.
"""
import os
import tempfile

from . import channel_mesh

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESHES = os.path.join(REPO, "tests", "golden", "meshes")

PHYSICS = {
    "pp": dict(kw=dict(), ref_cfg="pi_pp", text="PP mixing, no GM/Redi, no surface forcing"),
    "default": dict(kw=dict(mix_scheme="KPP", Fer_GM=True, Redi=True), ref_cfg="pi_default",
                    text="KPP mixing + GM + Redi (namelist.oce defaults), analytic wind/heat/fresh-water forcing"),
}


class Workload:
    """mesh directory + what is needed to start a run on it (global or rank-local mesh)"""

    def __init__(self, name, meshdir, mesh_kw, param_kw, dt, text, physics=None, levels=0, layers=47):
        self.name, self.meshdir, self.mesh_kw, self.param_kw, self.dt, self.text = name, meshdir, mesh_kw, param_kw, dt, text
        self.physics, self.levels, self.layers = physics, levels, layers

    def load_mesh(self, **kw):
        """the mesh of this workload.  The channel's initial state REDEFINES the Coriolis parameter of the mesh
        (toy_channel_soufflet.F90:303-307), so it is formed here, before any core copies the mesh to the device."""
        from .mesh import Mesh
        mesh = Mesh.load(self.meshdir, **dict(self.mesh_kw, **kw))
        if self.name == "channel":
            from . import toy_soufflet
            st = mesh.initial_state(2)
            aux = toy_soufflet.initial_state(mesh, st)
            mesh._channel_state = (st, aux)
        return mesh

    def params(self, **kw):
        from .config import make_params
        return make_params(**dict(self.param_kw, **kw))

    def initial_state(self, mesh):
        """returns (state, aux) -- aux: fields to set after upload (toy relaxation targets), forcing: dict or None"""
        if self.name == "channel":
            st, aux = mesh._channel_state
            return st, aux, None
        st = mesh.initial_state(2)
        from .synthetic import analytic_ts, analytic_forcing
        T, S = analytic_ts(self.meshdir)
        ln = mesh.myList_nod2D - 1
        st.tr_arr[0], st.tr_arr[1] = T[ln], S[ln]
        st.tr_arr_old[...] = st.tr_arr
        return st, {}, (analytic_forcing(mesh) if self.physics == "default" else None)

    def start(self, core, mesh):
        """upload the initial state (+ relaxation targets, forcing) into an OceanCore"""
        st, aux, forcing = self.initial_state(mesh)
        core.upload_state(st)
        for k, v in aux.items():
            core.set(k, v)
        if forcing:
            core.set_forcing(**forcing)
        if self.name == "channel":
            core.call("compute_zonal_mean")
        return st


def pi(physics="default", refine=0):
    d = os.path.join(MESHES, "pi")
    if refine > 0:                      # supplementary: pi refined uniformly (unstable beyond ~40 steps at 3 levels)
        from . import mesh_refine
        d = os.path.join(tempfile.gettempdir(), f"fesom_pi_refined_{refine}_{os.getpid()}")
        mesh_refine.refine(os.path.join(MESHES, "pi"), d, refine)
    return Workload("pi", d, dict(dt=900.0), dict(dt=900.0, **PHYSICS[physics]["kw"]), 900.0,
                    "pi mesh" + (f" refined {refine}x" if refine else "") + ", T/S tracers, zstar ALE, JM EOS, MFCT/QR4C/FCT advection, no sea ice, " + PHYSICS[physics]["text"],
                    physics=physics, levels=refine)


def basin(levels=3, layers=47, workdir=None):
    """BASELINE config #3 in kind with the reference's DEFAULT physics: the channel geometry refined `levels` times (3 -> 182 600 nodes) with an
    analytic bathymetry (continental slopes, a ridge, seamounts: ragged bottom levels, partial cells; levels by the reference partitioner's
    rule), Jackett-McDougall EOS, KPP + GM + Redi, analytic T/S and surface forcing as on pi -- setups/core2/setup.yml:7-12 keeps
    config/namelist.oce as it is.  No toy hooks.  The reference runs it too (oracle/ref/run_ref.py:basin_case)."""
    d = os.path.join(workdir or tempfile.gettempdir(), f"fesom_basin_r{levels}_{layers}")
    if not os.path.exists(os.path.join(d, "nlvls.out")):
        tmp = d + f".tmp{os.getpid()}"
        channel_mesh.build_basin(os.path.join(MESHES, "soufflet"), tmp, levels, layers)
        try:
            os.rename(tmp, d)
        except OSError:                 # another rank got there first
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    dt = channel_mesh.dt_for(levels)
    return Workload("basin", d, channel_mesh.basin_mesh_kw(levels), channel_mesh.basin_param_kw(levels), dt,
                    f"channel basin (Soufflet channel geometry refined {levels}x, analytic bathymetry with slopes / ridge / seamounts, partial cells), {layers} layers, T/S tracers, "
                    f"zstar ALE, JM EOS, KPP + GM + Redi (namelist.oce defaults), MFCT/QR4C/FCT advection, analytic wind/heat/fresh-water forcing, dt = {dt:g} s",
                    physics="default", levels=levels, layers=layers)


def channel(levels=3, layers=47, workdir=None):
    d = os.path.join(workdir or tempfile.gettempdir(), f"fesom_chan_r{levels}_{layers}")
    if not os.path.exists(os.path.join(d, "nlvls.out")):
        tmp = d + f".tmp{os.getpid()}"
        channel_mesh.build(os.path.join(MESHES, "soufflet"), tmp, levels, layers)
        try:
            os.rename(tmp, d)
        except OSError:                 # another rank got there first
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    dt = channel_mesh.dt_for(levels)
    return Workload("channel", d, channel_mesh.mesh_kw(levels), channel_mesh.param_kw(levels), dt,
                    f"Soufflet channel (reference CI case test_souf) refined {levels}x, {layers} layers, T/S tracers, zstar ALE, linear EOS, PP mixing, "
                    f"MFCT/QR4C/FCT advection, zonal relaxation hooks, dt = {dt:g} s", levels=levels, layers=layers)
