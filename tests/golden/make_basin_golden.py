#!/usr/bin/env python3
"""Golden numbers for the default-physics CORE2-class workload (fesom2_amd.workloads.basin: channel geometry refined 3x = 182 600 nodes,
analytic bathymetry, JM EOS, KPP + GM + Redi, analytic forcing): the eta extrema the REFERENCE prints (write_step_info,
src/write_step_info.F90) when oracle/_ref/fesom_oracle.x runs the mesh on 8 MPI ranks.  Needs /root/reference (build) -- run in the build
container: python tests/golden/make_basin_golden.py [NSTEPS].  Output: tests/golden/basin_r3_reference.json"""
import json, os, re, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)
from oracle.ref import run_ref

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
name, d = run_ref.basin_case(3, 8, 47)
rd = os.path.join(run_ref.OUT, f"run_{name}_8")
log = os.path.join(rd, "stdout.log")
if not (os.path.exists(log) and "--reuse" in sys.argv):
    rd, rc, lines = run_ref.run(name, 8, nsteps, mode="step", dump=(), dump_mesh=False, step_info=True)
    assert rc == 0
txt = open(log).read()
num = r"(-?\d*\.\d+(?:E[+-]\d+)?)"
mm = [tuple(float(x) for x in m.groups()) for m in re.finditer(r"min\(eta\) , max\(eta\)\s*=\s*" + num + r"\s+" + num, txt)]
tl = [l for l in txt.splitlines() if l.startswith("ORACLE_TIMING")]
keep = [1, 2, 3, 4, 5, 10, 20, 30, 40, 60, 80, 100, 150, 200, 240]
nod2D = int(open(os.path.join(d, "nod2d.out")).readline().split()[0])
out = {"source": "reference (oracle/_ref/fesom_oracle.x = the reference's own sources), 8 MPI ranks, channel basin refined 3x (analytic bathymetry), 47 layers, dt = 150 s, "
                 "JM EOS, KPP + GM + Redi, analytic forcing, write_step_info every step",
       "nod2D": nod2D, "steps_run": len(mm), "timing": tl[0] if tl else None, "eta_minmax": {str(k): list(mm[k - 1]) for k in keep if k <= len(mm)}}
json.dump(out, open(os.path.join(HERE, "basin_r3_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
