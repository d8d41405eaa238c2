#!/usr/bin/env python3
"""Golden vectors for the classic EVP rheology (whichEVP = 0, the default of the shipped namelist.ice): the REFERENCE's own EVPdynamics (src/ice_EVP.F90:397-667:
ice strength and sea-surface-slope term, per subcycle stress_tensor :23-134, stress2rhs :323-396, the node update) run by the harness driver (mode 'ice' with
ice_evp0, analytic ice state: oracle/ref/driver.F90:ice_harness) on the pi mesh: ONE MPI rank, two calls of 120 subcycles; and TWO ranks (dist_2), one call, rank-local
inputs and outputs -- the pin of the partitioned GPU path.  Needs /root/reference (build):  python tests/golden/make_ice_evp0_golden.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO); sys.path.insert(0, HERE)
from oracle.ref import run_ref
from refdump import read_dump

rd, rc, lines = run_ref.run("pi_pp", 1, 2, mode="ice", dump=(1, 2), ice_evp0=True)
assert rc == 0, open(os.path.join(rd, "stdout.log")).read()[-2000:]
out = {}
a = read_dump(os.path.join(rd, "dumps", "ice_in.r00000.bin"))
for k, v in a.items():
    out["in/" + k] = np.array(v)
for n in (1, 2):
    b = read_dump(os.path.join(rd, "dumps", f"ice_out{n:04d}.r00000.bin"))
    for k in ("u_ice", "v_ice", "sigma11", "sigma12", "sigma22"):
        out[f"out{n}/{k}"] = np.array(b[k])
rd2, rc2, lines2 = run_ref.run("pi_pp", 2, 1, mode="ice", dump=(1,), ice_evp0=True)
assert rc2 == 0
for r in range(2):
    a = read_dump(os.path.join(rd2, "dumps", f"ice_in.r{r:05d}.bin")); b = read_dump(os.path.join(rd2, "dumps", f"ice_out0001.r{r:05d}.bin"))
    for k in ("u_ice", "v_ice", "a_ice", "m_ice", "m_snow", "elevation", "u_w", "v_w", "stress_atmice_x", "stress_atmice_y", "sigma11", "sigma12", "sigma22", "metric_factor"):
        out[f"r2/{r}/in/{k}"] = np.array(a[k])
    for k in ("u_ice", "v_ice", "sigma11", "sigma12", "sigma22"):
        out[f"r2/{r}/out1/{k}"] = np.array(b[k])
np.savez_compressed(os.path.join(HERE, "ice_evp0_reference.npz"), **out)
print("wrote ice_evp0_reference.npz", lines, float(np.abs(out["out2/u_ice"]).max()))
