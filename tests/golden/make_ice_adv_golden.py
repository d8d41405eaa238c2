#!/usr/bin/env python3
"""Golden vectors for the sea-ice FCT advection: the REFERENCE's own ice_TG_rhs_div, ice_fct_solve (ice_solve_high_order, ice_solve_low_order,
ice_fem_fct x 3), ice_update_for_div (src/ice_fct.F90) and cut_off (src/ice_thermo_oce.F90:2-63), compiled into oracle/_ref/fesom_oracle.x and run by
the harness driver after every EVPdynamics_m call (mode 'ice' with ice_adv, the "Advection part" of ice_timestep, src/ice_setup_step.F90:213-232) on
the pi mesh: ONE rank, three calls, every intermediate array of calls 1 and 3 in full (3140 nodes); and TWO ranks (dist_2), one call, rank-local.
Needs /root/reference (build):  python tests/golden/make_ice_adv_golden.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
from oracle.ref import run_ref
from refdump import read_dump

ADV = ("evp.u_ice", "evp.v_ice", "tg.rhs_m", "tg.rhs_a", "tg.rhs_ms", "tg.rhs_mdiv", "tg.rhs_adiv", "tg.rhs_msdiv", "fct.m_icel", "fct.a_icel", "fct.m_snowl",
       "fct.dm_ice", "fct.da_ice", "fct.dm_snow", "fct.m_ice", "fct.a_ice", "fct.m_snow", "div.m_ice", "div.a_ice", "div.m_snow")
rd, rc, lines = run_ref.run("pi_pp", 1, 3, mode="ice", dump=(1, 2, 3), ice_adv=True)
assert rc == 0, open(os.path.join(rd, "stdout.log")).read()[-2000:]
out = {}
a = read_dump(os.path.join(rd, "dumps", "ice_in.r00000.bin"))
for k, v in a.items():
    out["in/" + k] = np.array(v)
for n in (1, 2, 3):
    b = read_dump(os.path.join(rd, "dumps", f"ice_adv{n:04d}.r00000.bin"))
    for k in ADV:
        out[f"adv{n}/{k}"] = np.array(b[k])
    b = read_dump(os.path.join(rd, "dumps", f"ice_out{n:04d}.r00000.bin"))
    for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "sigma11", "sigma12", "sigma22"):
        out[f"out{n}/{k}"] = np.array(b[k])
rd2, rc2, lines2 = run_ref.run("pi_pp", 2, 1, mode="ice", dump=(1,), ice_adv=True)
assert rc2 == 0
for r in range(2):
    a = read_dump(os.path.join(rd2, "dumps", f"ice_in.r{r:05d}.bin")); b = read_dump(os.path.join(rd2, "dumps", f"ice_out0001.r{r:05d}.bin"))
    for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "elevation", "u_w", "v_w", "stress_atmice_x", "stress_atmice_y", "sigma11", "sigma12", "sigma22", "metric_factor"):
        out[f"r2/{r}/in/{k}"] = np.array(a[k])
    for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow"):
        out[f"r2/{r}/out1/{k}"] = np.array(b[k])
np.savez_compressed(os.path.join(HERE, "ice_adv_reference.npz"), **out)
print("wrote ice_adv_reference.npz:", len(out), "arrays", lines)
