#!/usr/bin/env python3
"""GPU vs oracle, routine by routine, on the channel workload: prints every field that differs.  usage: chan_chain.py LEVELS [NSTEPS]"""
import os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from test_gpu_channel import start_pair, toy_chain
from parity_chain import compare
L = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl, mesh, gpu, orc = start_pair(L, tempfile.gettempdir())
print("mesh", mesh.nod2D, flush=True)
nbad = 0
for step in range(1, K + 1):
    for routine, arg, fields in toy_chain():
        gpu.call(routine, arg); orc.call(routine, arg)
        for f in fields:
            a, b = gpu.get(f, orc.count(f)), orc.get(f)
            ok, msg = compare(f, a, b)
            if not ok:
                nbad += 1
                bad = np.flatnonzero(a != b)
                print(f"step {step} {routine}({arg}) {msg}; first bad flat index {bad[:5]}", flush=True)
        if routine == "solve_ssh":
            print("   solver its gpu/orc", gpu.solver_iterations, orc.solver_iterations, flush=True)
    print("step", step, "done; mismatches so far", nbad, flush=True)
gpu.close()
