"""Write a partition of a mesh directory in the reference's own file format (`dist_<npes>/`: rpart.out, my_listNNNNN.out,
com_infoNNNNN.out; writer in the reference: src/oce_local.F90:162-317, readers: src/oce_mesh.F90:199-256,568-663 and
src/oce_ale.F90:1317-1344), from the host mesh layer's partition (csrc/mesh_host.cpp: node ownership from an existing
dist_<k*npes>, or a recursive coordinate bisection; owned/halo lists and communication lists by the reference's rules).
With it the REFERENCE can run on any mesh this package can load (e.g. the refined meshes of mesh_refine.py), and a FESOM2
user gets partitions without METIS."""
import ctypes as C
import os
import numpy as np

from .mesh import Mesh


def _wrap(vals, per=6):
    vals = [int(v) for v in vals]
    return "\n".join(" ".join(f"{v:11d}" for v in vals[i:i + per]) for i in range(0, len(vals), per))


def _arr(p, n):
    return np.ctypeslib.as_array(p, shape=(n,)).copy() if n > 0 else np.zeros(0, dtype=np.int32)


def write_dist(meshdir, npes, outdir=None, **mesh_kw):
    """returns the directory written (default <meshdir>/dist_<npes>)"""
    outdir = outdir or os.path.join(meshdir, f"dist_{npes}")
    os.makedirs(outdir, exist_ok=True)
    owned = []
    for r in range(npes):
        m = Mesh.load(meshdir, npes=npes, mype=r, **mesh_kw)
        myN, eN = m.myDim_nod2D, m.eDim_nod2D
        d = m.d
        myE, eE, eX, myD, eD = d.myDim_elem2D, d.eDim_elem2D, d.eXDim_elem2D, d.myDim_edge2D, d.eDim_edge2D
        owned.append(m.myList_nod2D[:myN].copy())
        with open(os.path.join(outdir, f"my_list{r:05d}.out"), "w") as f:
            f.write(f"{r:12d}\n{myN:12d}\n{eN:12d}\n{_wrap(m.myList_nod2D[:myN + eN])}\n")
            f.write(f"{myE:12d}\n{eE:12d}\n{eX:12d}\n{_wrap(m.myList_elem2D[:myE + eE + eX])}\n")
            f.write(f"{myD:12d}\n{eD:12d}\n{_wrap(m.myList_edge2D[:myD + eD])}\n")
        part = m.part_p.contents
        with open(os.path.join(outdir, f"com_info{r:05d}.out"), "w") as f:
            f.write(f"{r:12d}\n")
            for c in (part.com_nod2D, part.com_elem2D, part.com_elem2D_full):
                for pe_n, pe, ptr, lst in ((c.rPEnum, c.rPE, c.rptr, c.rlist), (c.sPEnum, c.sPE, c.sptr, c.slist)):
                    p = _arr(ptr, pe_n + 1) if pe_n > 0 else np.array([1], dtype=np.int32)
                    f.write(f"{pe_n:12d}\n{_wrap(_arr(pe, pe_n))}\n{_wrap(p)}\n{_wrap(_arr(lst, int(p[-1]) - 1))}\n")
        m.close() if hasattr(m, "close") else None
    # rpart.out: number of ranks, owned-node counts, then for every node (global order) its index in the rank-contiguous numbering
    nod2D = int(sum(len(o) for o in owned))
    mapping = np.zeros(nod2D, dtype=np.int64)
    off = 0
    for o in owned:
        mapping[o - 1] = off + np.arange(1, len(o) + 1)
        off += len(o)
    with open(os.path.join(outdir, "rpart.out"), "w") as f:
        f.write(f"{npes:12d}\n{_wrap([len(o) for o in owned], per=8)}\n")
        f.write("\n".join(f"{int(v):12d}" for v in mapping) + "\n")
    return outdir


def write_edge_files(meshdir, **mesh_kw):
    """edgenum.out / edges.out / edge_tri.out of a mesh directory that lacks them (refined meshes), in the format the reference
    reads (src/oce_mesh.F90:1457-1520): the host mesh layer generates the edges by the reference partitioner's rule
    (find_edges_ini, src/fvom_init.F90:315-650; csrc/mesh_host.cpp:generate_edges)."""
    m = Mesh.load(meshdir, **mesh_kw)
    D, Din = m.d.edge2D, m.d.edge2D_in
    with open(os.path.join(meshdir, "edgenum.out"), "w") as f:
        f.write(f"{D:12d}\n{Din:12d}\n")
    np.savetxt(os.path.join(meshdir, "edges.out"), m.edges[:D], fmt="%11d")
    et = m.edge_tri[:D].copy()
    et[et[:, 1] <= 0, 1] = -999
    np.savetxt(os.path.join(meshdir, "edge_tri.out"), et, fmt="%11d")
    return D, Din
