#!/usr/bin/env python3
"""Step time of the Fortran drop-in executable (oracle/_ref/fesom_gpu_dropin.x: the reference's own set-up + the Fortran host
layer + libfesom_gpu.so): forcing uploaded every step, status_check after every call; 1 rank, and 2 MPI ranks sharing the GPU
(host-staged MPI halo transport).  Prints the executable's own timing lines.  usage: dropin_timing.py [nsteps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.ref import run_ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
os.environ["FESOM_GPU_DEVICE"] = "0"
for cfg in ("pi_pp", "pi_default"):
    for ranks in (1, 2):
        rd, rc, lines = run_ref.run(cfg, ranks, n, mode="gpu", dump=(), dump_mesh=False, exe_name="fesom_gpu_dropin.x")
        print(cfg, "gpu ranks", ranks, "rc", rc, [l for l in lines if "TIMING" in l], flush=True)
    rd, rc, lines = run_ref.run(cfg, 8, n, mode="step", dump=(), dump_mesh=False)
    print(cfg, "cpu ranks 8 rc", rc, [l for l in lines if "TIMING" in l], flush=True)
