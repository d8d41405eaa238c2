#!/usr/bin/env python3
"""Golden vectors for the adaptive EVP rheology (whichEVP = 2): the REFERENCE's own EVPdynamics_a (src/ice_maEVP.F90:785-888: ssh2rhs, stress_tensor_a,
stress2rhs_m, the node update with beta_evp_array, find_alpha_field_a, find_beta_field_a) run by the harness driver (mode 'ice' with ice_aevp, analytic ice
state: oracle/ref/driver.F90:ice_harness) on the pi mesh with ONE MPI rank, three calls of 120 subcycles (alpha / beta adapt after every call).  Inputs and
outputs in full.  Needs /root/reference (build):  python tests/golden/make_ice_aevp_golden.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
from oracle.ref import run_ref
from refdump import read_dump

rd, rc, lines = run_ref.run("pi_pp", 1, 3, mode="ice", dump=(1, 2, 3), ice_aevp=True)
assert rc == 0, open(os.path.join(rd, "stdout.log")).read()[-2000:]
out = {}
a = read_dump(os.path.join(rd, "dumps", "ice_in.r00000.bin"))
for k, v in a.items():
    out["in/" + k] = np.array(v)
for n in (1, 2, 3):
    b = read_dump(os.path.join(rd, "dumps", f"ice_out{n:04d}.r00000.bin"))
    for k in ("u_ice", "v_ice", "sigma11", "sigma12", "sigma22", "alpha_evp_array", "beta_evp_array"):
        out[f"out{n}/{k}"] = np.array(b[k])
# the same on TWO ranks (dist_2), one call: rank-local inputs and outputs -- the pin of the partitioned GPU path
rd2, rc2, lines2 = run_ref.run("pi_pp", 2, 1, mode="ice", dump=(1,), ice_aevp=True)
assert rc2 == 0
for r in range(2):
    a = read_dump(os.path.join(rd2, "dumps", f"ice_in.r{r:05d}.bin")); b = read_dump(os.path.join(rd2, "dumps", f"ice_out0001.r{r:05d}.bin"))
    for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "elevation", "u_w", "v_w", "stress_atmice_x", "stress_atmice_y", "sigma11", "sigma12", "sigma22", "metric_factor",
              "alpha_evp_array", "beta_evp_array"):
        out[f"r2/{r}/in/{k}"] = np.array(a[k])
    for k in ("u_ice", "v_ice", "sigma11", "sigma12", "sigma22", "alpha_evp_array", "beta_evp_array"):
        out[f"r2/{r}/out1/{k}"] = np.array(b[k])
np.savez_compressed(os.path.join(HERE, "ice_aevp_reference.npz"), **out)
print("wrote ice_aevp_reference.npz:", {k: v.shape for k, v in out.items() if k.startswith("out1/")}, lines)
