#!/usr/bin/env python3
"""Synthetic ice-shelf cavity on the pi mesh (the reference ships no cavity mesh): writes tests/golden/meshes/pi_cavity = the pi mesh files (the reference's
test fixture, as data) + the three files the reference reads with use_cavity=.true. (src/oce_mesh.F90:897-1280: cavity_elvls.out, cavity_nlvls.out,
cavity_depth.out).  The upper levels are derived from an analytic cavity draft (a cap of up to 260 m over the deep part of a lon/lat box) by the rule of the
reference's partitioner, restated here (src/fvom_init.F90:878-1209 find_levels_cavity: first mid-layer depth below the mean draft of the element, at least three
open layers, every open cell with two open neighbours, consistency between element and node levels), so that the reference accepts the geometry.
usage: make_cavity_mesh.py [OUTDIR]"""
import os
import shutil
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "meshes", "pi")


def read_mesh(d):
    with open(os.path.join(d, "nod2d.out")) as f:
        n = int(f.readline())
        xy = np.array([[float(t) for t in f.readline().split()[1:3]] for _ in range(n)])
    with open(os.path.join(d, "elem2d.out")) as f:
        ne = int(f.readline())
        el = np.array([[int(t) for t in f.readline().split()] for _ in range(ne)], dtype=np.int64) - 1
    with open(os.path.join(d, "aux3d.out")) as f:
        nl = int(f.readline())
        zbar = np.array([float(f.readline()) for _ in range(nl)])
    nlev_e = np.loadtxt(os.path.join(d, "elvls.out"), dtype=np.int64)
    nlev_n = np.loadtxt(os.path.join(d, "nlvls.out"), dtype=np.int64)
    return xy, el, zbar, nlev_e, nlev_n


def neighbours(el, nn):
    """elements sharing an edge with each element (-1: none), as elem_neighbors of the partitioner"""
    edge = {}
    nb = -np.ones((len(el), 3), dtype=np.int64)
    for e, tri in enumerate(el):
        for j in range(3):
            key = tuple(sorted((tri[j], tri[(j + 1) % 3])))
            if key in edge:
                o, oj = edge[key]
                nb[e, j] = o; nb[o, oj] = e
            else:
                edge[key] = (e, j)
    return nb


def find_levels_cavity(el, nb, nlev_e, nlev_n, Z, draft, nl):
    """src/fvom_init.F90:878-1209 (levels 1-based as there)"""
    E, N = len(el), len(nlev_n)
    ulev = np.ones(E, dtype=np.int64)
    for e in range(E):
        dmean = draft[el[e]].sum() / 3.0                       # which_depth_n2e = 'mean'
        if dmean < 0.0:
            ulev[e] = 2
        for nz in range(1, nlev_e[e]):
            if Z[nz - 1] < dmean or nlev_e[e] - nz <= 3:
                ulev[e] = nz
                break
    maxlev = int(ulev.max())
    fix = np.zeros(E, dtype=bool)
    nie = [[] for _ in range(N)]
    for e in range(E):
        for n in el[e]:
            nie[n].append(e)
    for outer in range(10):
        red = np.zeros(E, dtype=bool)
        for nz in range(1, maxlev + 1):
            for it in range(1000):
                done = True
                for e in range(E):
                    if nz >= ulev[e] and nz < nlev_e[e]:
                        cnt = sum(1 for o in nb[e] if o >= 0 and ulev[o] <= nz and nlev_e[o] > nz)
                        if cnt < 2:
                            if nlev_e[e] - (nz + 1) >= 3 and not red[e] and not fix[e]:
                                ulev[e] = nz + 1
                            else:
                                cand = [(ulev[o] - nz, k) for k, o in enumerate(nb[e]) if o >= 0 and ulev[o] - nz > 0]
                                o = nb[e][min(cand)[1]]
                                ulev[o] = nz - 1; red[o] = True
                            done = False
                if done:
                    break
        uln = np.full(N, nl, dtype=np.int64)
        for e in range(E):
            for n in el[e]:
                uln[n] = min(uln[n], ulev[e])
        ok = True
        if (ulev >= nlev_e).any() or (nlev_e - ulev < 3).any() or (uln >= nlev_n).any() or (nlev_n - uln < 3).any():
            ok = False
        if any(ulev[e] < uln[el[e]].max() for e in range(E)):
            ok = False
        for n in range(N):
            num = np.zeros(nl + 2, dtype=np.int64)
            for e in nie[n]:
                num[ulev[e]:nlev_e[e]] += 1
            for nz in range(uln[n], nlev_n[n]):
                if num[nz] == 0:
                    ok = False
                    for e in nie[n]:
                        if ulev[e] > nz:
                            ulev[e] = nz; fix[e] = True
                elif num[nz] == 1:
                    ok = False
        if ok:
            return ulev, uln
    raise RuntimeError("cavity geometry did not converge")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "meshes", "pi_cavity")
    xy, el, zbar, nlev_e, nlev_n = read_mesh(SRC)
    Z = 0.5 * (zbar[:-1] + zbar[1:])                           # mid-layer depths (negative)
    lon, lat = xy[:, 0], xy[:, 1]
    # the draft: a smooth cap over the deep part of a box; 0 elsewhere (open ocean)
    depth_n = -zbar[nlev_n - 1]                                # bottom depth of the node's column
    lo0, lo1 = np.percentile(lon, 25), np.percentile(lon, 60)
    la0, la1 = np.percentile(lat, 30), np.percentile(lat, 65)
    sx = np.clip(np.minimum(lon - lo0, lo1 - lon) / (0.25 * (lo1 - lo0)), 0.0, 1.0)
    sy = np.clip(np.minimum(lat - la0, la1 - lat) / (0.25 * (la1 - la0)), 0.0, 1.0)
    draft = -260.0 * sx * sy * (depth_n > 1500.0)
    nb = neighbours(el, len(lon))
    ulev, uln = find_levels_cavity(el, nb, nlev_e, nlev_n, Z, draft, len(zbar))
    if os.path.isdir(out):
        shutil.rmtree(out)
    shutil.copytree(SRC, out)
    np.savetxt(os.path.join(out, "cavity_elvls.out"), ulev, fmt="%d")
    np.savetxt(os.path.join(out, "cavity_nlvls.out"), uln, fmt="%d")
    np.savetxt(os.path.join(out, "cavity_depth.out"), np.rint(draft).astype(np.int64), fmt="%d")
    print("cavity elements", int((ulev > 1).sum()), "of", len(ulev), "max upper level", int(ulev.max()), "nodes under the shelf", int((uln > 1).sum()))


if __name__ == "__main__":
    main()
