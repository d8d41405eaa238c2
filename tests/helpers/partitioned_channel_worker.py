"""Worker of tests/test_gpu_partitioned.py::test_partitioned_channel (one process per rank, gloo rendezvous, ranks share GPU 0):
the CORE2-class workload's mesh family -- the Soufflet channel refined CHAN_LEVELS times, with its toy hooks (global zonal means
every 10th step, zonal relaxation) -- partitioned over the ranks against the single-partition run of the same steps."""
import json, os, sys
import numpy as np
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from fesom2_amd import workloads, parallel
from fesom2_amd.core import OceanCore

MAKE = workloads.basin if os.environ.get("CHAN_WORKLOAD") == "basin" else workloads.channel      # basin: the same geometry with bathymetry and the default physics
LEVELS = int(os.environ.get("CHAN_LEVELS", "1"))
NSTEPS = int(os.environ.get("CHAN_NSTEPS", "12"))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == 0:
        MAKE(LEVELS, workdir=os.environ.get("CHAN_WORKDIR"))
    dist.barrier()
    wl = MAKE(LEVELS, workdir=os.environ.get("CHAN_WORKDIR"))
    gm = wl.load_mesh()
    g = OceanCore(gm, wl.params()); wl.start(g, gm)
    g.run_steps(1, NSTEPS)
    n1 = gm.nl - 1
    ref = {"eta_n": g.get("eta_n", gm.nod2D), "tr_arr": g.get("tr_arr", 2 * gm.nod2D * n1).reshape(2, gm.nod2D, n1), "UV": g.get("UV", 2 * n1 * gm.elem2D).reshape(gm.elem2D, n1, 2)}
    its_single = g.solver_iterations
    g.close()
    pc = parallel.PartitionedCore(wl, transport=os.environ.get("PART_TRANSPORT") or None)
    for n in range(1, NSTEPS + 1):
        pc.step_native(n)
    lm = pc.mesh
    myN, myE = lm.myDim_nod2D, lm.myDim_elem2D
    ln, le = lm.myList_nod2D[:myN] - 1, lm.myList_elem2D[:myE] - 1
    N, E = myN + lm.eDim_nod2D, myE + lm.eDim_elem2D
    eta = pc.core.get("eta_n", N)[:myN]
    T = pc.core.get("tr_arr", 2 * N * n1).reshape(2, N, n1)[:, :myN]
    UV = pc.core.get("UV", 2 * n1 * E).reshape(E, n1, 2)[:myE]
    rep = {"rank": rank, "d_eta": float(np.abs(eta - ref["eta_n"][ln]).max()), "d_T": float(np.abs(T - ref["tr_arr"][:, ln]).max()),
           "d_UV": float(np.abs(UV - ref["UV"][le]).max()), "iters": [int(its_single), int(pc.solver_iterations)], "owned": int(myN),
           "transport": pc.transport_name, "eta_range": [float(eta.min()), float(eta.max())]}
    pc.close()
    sys.stdout.write("CHANREPORT " + json.dumps(rep) + chr(10)); sys.stdout.flush()
    dist.destroy_process_group()


main()
