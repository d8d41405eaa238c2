"""Uniform refinement of a FESOM mesh directory (every triangle -> 4 by its edge midpoints), written in the reference's ASCII
mesh format (nod2d.out, elem2d.out, aux3d.out, elvls.out, nlvls.out).  Used to obtain production-size meshes (CORE2 class and
beyond) from the small test meshes that ship with the reference, for measurements and for the partitioned path; the edge files
are left out on purpose: the host mesh layer generates them by the reference partitioner's own rule
(csrc/mesh_host.cpp:generate_edges = find_edges_ini).  Children inherit the parent's number of levels, a midpoint node the
mean depth of its edge; nodes are renumbered along the element order so that neighbours stay close in memory."""
import os
import numpy as np


def _read(meshdir):
    t = open(os.path.join(meshdir, "nod2d.out")).read().split()
    n = int(t[0]); a = np.array(t[1:1 + 4 * n], dtype=np.float64).reshape(n, 4)
    lon, lat, flag = a[:, 1].copy(), a[:, 2].copy(), a[:, 3].astype(np.int64)
    t = open(os.path.join(meshdir, "elem2d.out")).read().split()
    e = int(t[0]); el = np.array(t[1:1 + 3 * e], dtype=np.int64).reshape(e, 3) - 1
    t = open(os.path.join(meshdir, "aux3d.out")).read().split()
    nl = int(t[0]); zbar = [x for x in t[1:1 + nl]]; depth = np.array(t[1 + nl:1 + nl + n], dtype=np.float64)
    elv = np.array(open(os.path.join(meshdir, "elvls.out")).read().split(), dtype=np.int64)
    return lon, lat, flag, el, nl, zbar, depth, elv


def refine_once(lon, lat, flag, el, depth, elv, cyclic_deg=360.0):
    n = lon.size
    pairs = np.concatenate([el[:, [0, 1]], el[:, [1, 2]], el[:, [2, 0]]])            # edge k of element e at row k*E + e
    lo, hi = pairs.min(1), pairs.max(1)
    key = lo * n + hi
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    a, b = uniq // n, uniq % n
    l1, l2 = lon[a], lon[b].copy()
    d = l2 - l1
    l2[d > cyclic_deg / 2] -= cyclic_deg; l2[d < -cyclic_deg / 2] += cyclic_deg
    mlon, mlat = 0.5 * (l1 + l2), 0.5 * (lat[a] + lat[b])
    mflag = (cnt == 1).astype(np.int64)                                              # midpoint of a boundary edge
    mdepth = 0.5 * (depth[a] + depth[b])
    E = el.shape[0]
    m12, m23, m31 = n + inv[:E], n + inv[E:2 * E], n + inv[2 * E:]
    n1, n2, n3 = el[:, 0], el[:, 1], el[:, 2]
    child = np.stack([np.stack([n1, m12, m31], 1), np.stack([m12, n2, m23], 1), np.stack([m31, m23, n3], 1), np.stack([m12, m23, m31], 1)], 1)
    nel = child.reshape(4 * E, 3)
    nlon, nlat = np.concatenate([lon, mlon]), np.concatenate([lat, mlat])
    nflag, ndepth = np.concatenate([flag, mflag]), np.concatenate([depth, mdepth])
    nelv = np.repeat(elv, 4)
    # renumber nodes by the first element that contains them
    N = nlon.size
    first = np.full(N, 4 * E, dtype=np.int64)
    np.minimum.at(first, nel.ravel(), np.repeat(np.arange(4 * E), 3))
    order = np.argsort(first, kind="stable")
    new_id = np.empty(N, dtype=np.int64); new_id[order] = np.arange(N)
    return nlon[order], nlat[order], nflag[order], new_id[nel], ndepth[order], nelv


def refine(meshdir, outdir, levels=1, cyclic_deg=360.0):
    lon, lat, flag, el, nl, zbar, depth, elv = _read(meshdir)
    for _ in range(levels):
        lon, lat, flag, el, depth, elv = refine_once(lon, lat, flag, el, depth, elv, cyclic_deg)
    N, E = lon.size, el.shape[0]
    nlv = np.zeros(N, dtype=np.int64)
    np.maximum.at(nlv, el.ravel(), np.repeat(elv, 3))                               # nlevels_nod2D = max over the node's elements
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, "nod2d.out"), "w") as f:
        f.write(f"{N}\n")
        f.write("".join(f"{i + 1} {lon[i]:.10f} {lat[i]:.10f} {flag[i]}\n" for i in range(N)))
    with open(os.path.join(outdir, "elem2d.out"), "w") as f:
        f.write(f"{E}\n")
        np.savetxt(f, el + 1, fmt="%d")
    with open(os.path.join(outdir, "aux3d.out"), "w") as f:
        f.write(f"{nl}\n" + "\n".join(zbar) + "\n")
        np.savetxt(f, depth, fmt="%.6f")
    np.savetxt(os.path.join(outdir, "elvls.out"), elv, fmt="%d")
    np.savetxt(os.path.join(outdir, "nlvls.out"), nlv, fmt="%d")
    return N, E
