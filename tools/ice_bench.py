#!/usr/bin/env python3
"""Sea-ice mEVP rheology (EVPdynamics_m, 120 subcycles per call) on the GPU: device time per call on the pi mesh and on the CORE2-class
channel mesh (analytic ice state, harness formulas), next to the reference's own routine on the host cores (oracle/_ref, driver mode
'ice', pi, best of 1/8/16 MPI ranks).  usage: ice_bench.py [out.json]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from fesom2_amd import workloads, ice
from fesom2_amd.mesh import Mesh


def analytic_state(mesh):
    """the ice state of oracle/ref/driver.F90:ice_harness on any mesh"""
    d = mesh.desc_p.contents
    N, E = mesh.myDim_nod2D + mesh.eDim_nod2D, mesh.myDim_elem2D
    geo = np.ctypeslib.as_array(d.geo_coord_nod2D, shape=(N, 2))
    lon, lat = geo[:, 0], geo[:, 1]
    a = np.clip(0.55 + 0.6 * np.sin(2 * lat) + 0.25 * np.cos(3 * lon), 0.0, 1.0)
    a = np.where((a < 0.02) & (np.cos(5 * lon) > 0), 0.005, a)
    edges = np.ctypeslib.as_array(d.edges, shape=(mesh.myDim_edge2D, 2)); lst = np.ctypeslib.as_array(d.myList_edge2D, shape=(mesh.myDim_edge2D,))
    bc = np.ones(N); bc[edges[lst > d.edge2D_in].ravel() - 1] = 0
    return ice.IceFields(a_ice=a, m_ice=a * (1.2 + 0.9 * np.cos(2 * lon + 1.0)), m_snow=0.15 * a * (1 + np.sin(lon)), u_ice=0.08 * np.sin(lon) * np.cos(lat) * bc,
                         v_ice=0.05 * np.cos(2 * lon) * bc, u_w=0.12 * np.cos(lon + 0.5), v_w=0.07 * np.sin(2 * lat), elevation=0.3 * np.sin(2 * lon) * np.cos(lat),
                         stress_atmice_x=0.12 * np.cos(3 * lat), stress_atmice_y=0.05 * np.sin(2 * lon + 0.3), sigma11=np.zeros(E), sigma12=np.zeros(E), sigma22=np.zeros(E))


def gpu(mesh, dt):
    par = ice.ice_params(ice_dt=dt)
    core = ice.IceCore(mesh, par)
    st = analytic_state(mesh)
    core.upload(st)
    ms = core.time_ms(10)
    core.download(st)
    assert np.isfinite(st["u_ice"]).all()
    # whole ice steps (EVP + FCT advection incl. the host-side exp of the pressure factor): wall time over 10 steps, state finite afterwards
    core.step(2); core.download(st)
    t0 = time.perf_counter()
    core.step(10); core.download(st)
    step_ms = (time.perf_counter() - t0) * 1e3 / 10
    assert np.isfinite(st["m_ice"]).all() and st["a_ice"].max() <= 1.0
    core.close()
    E = mesh.myDim_elem2D; N = mesh.myDim_nod2D
    # algorithmic traffic of one subcycle: stresses read + written (6 E), gradients + metric + area (8 E), velocities read + written (4 N), node data (10 N)
    by = 8.0 * (14 * E + 14 * N)
    return {"nodes": int(N), "elements": int(E), "ms_per_call": round(ms, 4), "us_per_subcycle": round(ms * 1e3 / 120, 3), "subcycles": 120,
            "algorithmic_GBs": round(by * 120 / (ms * 1e-3) / 1e9, 1), "umax": float(np.abs(st["u_ice"]).max()),
            "ms_per_ice_step_evp_plus_advection_wall": round(step_ms, 4), "ms_advection_wall": round(step_ms - ms, 4)}


def gpu_rheology(mesh, dt, which):
    """device time per call of the classic (0) / adaptive (2) EVP on the analytic state"""
    par = ice.ice_params(ice_dt=dt, whichEVP=which)
    core = ice.IceCore(mesh, par)
    st = analytic_state(mesh)
    if which == 2:
        st = ice.IceFields(**dict(st.a, alpha_evp_array=np.full(mesh.myDim_elem2D, par.alpha_evp), beta_evp_array=np.full(mesh.myDim_nod2D + mesh.eDim_nod2D, par.alpha_evp)))
    core.upload(st)
    ms = core.time_ms(10)
    core.download(st)
    assert np.isfinite(st["u_ice"]).all()
    core.close()
    return {"ms_per_call": round(ms, 4), "us_per_subcycle": round(ms * 1e3 / 120, 3), "umax": float(np.abs(st["u_ice"]).max())}


def main():
    out = {"what": "EVPdynamics_m (src/ice_maEVP.F90:273-602), one call = 120 subcycles, fp64, analytic ice state"}
    pim = Mesh.load(os.path.join(REPO, "tests", "golden", "meshes", "pi"), dt=900.0)
    out["pi"] = gpu(pim, 900.0)
    wl = workloads.channel(3)
    out["channel_r3"] = gpu(wl.load_mesh(), wl.dt)
    out["other_rheologies"] = {"what": "whichEVP = 0: classic EVP, EVPdynamics (src/ice_EVP.F90:397-667, the default of namelist.ice); whichEVP = 2: adaptive EVP, EVPdynamics_a "
                                       "(src/ice_maEVP.F90:785-888); two launches per subcycle, thread per node / element"}
    for which, name in ((0, "classic_evp"), (2, "adaptive_evp")):
        out["other_rheologies"][name] = {"pi": gpu_rheology(pim, 900.0, which), "channel_r3": gpu_rheology(wl.load_mesh(), wl.dt, which)}
    try:
        from oracle.ref import run_ref
        for name, kw in (("classic_evp", dict(ice_evp0=True)), ("adaptive_evp", dict(ice_aevp=True))):
            tr = {}
            for ranks in (1, 8, 16):
                if ranks > (os.cpu_count() or 1):
                    continue
                rd, rc, lines = run_ref.run("pi_pp", ranks, 20, mode="ice", dump=(), **kw)
                tl = [l for l in lines if l.startswith("ORACLE_TIMING_ICE")]
                if rc == 0 and tl:
                    tr[ranks] = round(float(tl[0].split("s_per_call=")[1].split()[0]) * 1e3, 3)
            if tr:
                b = min(tr, key=tr.get)
                out["other_rheologies"][name]["cpu_reference_pi"] = {"ms_per_call_by_ranks": tr, "best_ranks": b, "gpu_over_cpu": round(tr[b] / out["other_rheologies"][name]["pi"]["ms_per_call"], 1)}
        tried, tried_adv = {}, {}
        for ranks in (1, 8, 16):
            if ranks > (os.cpu_count() or 1):
                continue
            rd, rc, lines = run_ref.run("pi_pp", ranks, 20, mode="ice", dump=())
            tl = [l for l in lines if l.startswith("ORACLE_TIMING_ICE")]
            if rc == 0 and tl:
                tried[ranks] = round(float(tl[0].split("s_per_call=")[1].split()[0]) * 1e3, 3)
            rd, rc, lines = run_ref.run("pi_pp", ranks, 20, mode="ice", dump=(), ice_adv=True)
            tl = [l for l in lines if l.startswith("ORACLE_TIMING_ICE")]
            if rc == 0 and tl:
                tried_adv[ranks] = round(float(tl[0].split("s_per_call=")[1].split()[0]) * 1e3, 3)
        best = min(tried, key=tried.get)
        out["cpu_reference_pi"] = {"ms_per_call_by_ranks": tried, "best_ranks": best, "ms_per_call": tried[best], "host_cores": os.cpu_count(),
                                   "gpu_over_cpu": round(tried[best] / out["pi"]["ms_per_call"], 1),
                                   "ms_per_ice_step_evp_plus_advection_by_ranks": tried_adv,
                                   "gpu_over_cpu_ice_step": round(min(tried_adv.values()) / out["pi"]["ms_per_ice_step_evp_plus_advection_wall"], 1) if tried_adv else None}
    except Exception as e:      # noqa: BLE001
        out["cpu_reference_pi"] = {"error": str(e)[:300]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


main()
