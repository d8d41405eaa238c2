#!/usr/bin/env python3
"""Diagnose the channel workload on the GPU: (a) eager 4-stream DAG vs single-stream run, bit for bit after K steps;
(b) step-by-step eta extrema (compare with the reference's write_step_info).  usage: chan_diag.py LEVELS NSTEPS"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
L, K = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:                       # child: run K steps, dump the state
    from fesom2_amd import workloads
    from fesom2_amd.core import OceanCore
    wl = workloads.channel(L)
    mesh = wl.load_mesh()
    core = OceanCore(mesh, wl.params())
    wl.start(core, mesh)
    n1 = mesh.nl - 1
    for n in range(1, K + 1):
        core.run_steps(n, 1)
        if n <= 5 or n % 10 == 0:
            e = core.get("eta_n", mesh.nod2D)
            print(sys.argv[3], n, "its", core.solver_iterations, "eta %.16e %.16e" % (e.min(), e.max()), flush=True)
    np.savez(sys.argv[3], eta=core.get("eta_n", mesh.nod2D), tr=core.get("tr_arr", 2 * n1 * mesh.nod2D), uv=core.get("UV", 2 * n1 * mesh.elem2D))
    core.close()
    sys.exit(0)
outs = {}
for tag, env in (("dag", {}), ("serial", {"FESOM_GPU_SERIAL": "1"})):
    fn = f"/tmp/chan_diag_{tag}.npz"
    subprocess.run([sys.executable, __file__, str(L), str(K), fn], env=dict(os.environ, **env), check=True)
    outs[tag] = np.load(fn)
for f in ("eta", "tr", "uv"):
    a, b = outs["dag"][f], outs["serial"][f]
    print(f, "dag vs serial: max abs diff", float(np.abs(a - b).max()), "bitwise", bool((a.view(np.int64) == b.view(np.int64)).all()))
