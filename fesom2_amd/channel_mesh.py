"""CORE2-class workload that the REFERENCE runs too: the Soufflet channel of the reference's own CI case
(setups/test_souf/setup.yml, src/toy_channel_soufflet.F90) refined uniformly and, optionally, re-layered.

The reference ships no CORE2 mesh (BASELINE config #3: ~127 k nodes, 47 levels).  The Soufflet channel is its stable,
analytically balanced case (zonal jet in thermal-wind balance + small perturbation, cyclic in x, flat bottom at 4000 m,
Soufflet et al. 2016 ran it from 20 km down to 2 km): every uniform refinement (mesh_refine.py, cyclic edges) quarters
the cells, three levels give 184 000 nodes at 2.5 km.  `layers=47` replaces the 40 stretched layers by 47 with the same
stretching law (dz_{k+1} = 1.1 dz_k, sum = 4000 m), i.e. the vertical extent of pi / CORE2.

The time step follows the horizontal resolution (setups/test_souf: 1200 s at 20 km): dt = 1200 s / 2**levels.
Everything the reference needs to run on the mesh (edge files, dist_<npes>) is written by partition_io in its formats.
"""
import os
import numpy as np

from . import mesh_refine

CYCLIC_DEG = 4.5               # setups/test_souf/setup.yml:33
BASE_DT = 1200.0               # step_per_day = 72 at 20 km


def stretched_zbar(layers, depth=4000.0, ratio=1.1):
    """layer interfaces (positive down, as in the channel's aux3d.out) of `layers` layers with dz_{k+1} = ratio*dz_k"""
    dz0 = depth * (ratio - 1.0) / (ratio ** layers - 1.0)
    z = np.concatenate([[0.0], np.cumsum(dz0 * ratio ** np.arange(layers))])
    z[-1] = depth
    return z


def build(base_dir, outdir, levels, layers=None):
    """refine `base_dir` (the soufflet test mesh) `levels` times into `outdir`; returns (nodes, elements, nl)"""
    N, E = mesh_refine.refine(base_dir, outdir, levels, cyclic_deg=CYCLIC_DEG)
    t = open(os.path.join(outdir, "aux3d.out")).read().split()
    nl = int(t[0])
    if layers is not None and layers + 1 != nl:
        depth = np.array(t[1 + nl:1 + nl + N], dtype=np.float64)
        assert np.all(depth == depth[0]), "re-layering is only defined for the flat-bottom channel"
        zbar = stretched_zbar(layers, abs(float(depth[0])))
        nl = layers + 1
        with open(os.path.join(outdir, "aux3d.out"), "w") as f:
            f.write(f"{nl}\n" + "\n".join(f"{z:.10g}" for z in zbar) + "\n")
            np.savetxt(f, depth, fmt="%.6f")
        np.savetxt(os.path.join(outdir, "elvls.out"), np.full(E, nl, dtype=np.int64), fmt="%d")
        np.savetxt(os.path.join(outdir, "nlvls.out"), np.full(N, nl, dtype=np.int64), fmt="%d")
    return N, E, nl


THERS_ZBAR_LEV = 5             # gen_modules_config.F90: shallowest allowed bottom level


def basin_depth(lon, lat):
    """analytic bathymetry of the `basin` variant of the channel [m, positive]: 4000 m abyss, continental slopes towards the southern
    and northern walls (shelf at ~350 m), a meridional ridge (1900 m high) across the cyclic direction, and a field of seamounts --
    ragged bottom levels and partial bottom cells everywhere, as on a real mesh"""
    y = lat / 18.0                                              # 0 .. 1 across the channel (lat0 = 0, 2000 km = 18 degrees)
    x = lon / CYCLIC_DEG                                        # 0 .. 1 along the cyclic direction
    slope = 3650.0 * (np.exp(-(y / 0.09) ** 2) + np.exp(-((1.0 - y) / 0.09) ** 2))
    dx = np.minimum(np.abs(x - 0.55), 1.0 - np.abs(x - 0.55))
    ridge = 1900.0 * np.exp(-(dx / 0.07) ** 2) * (0.6 + 0.4 * np.cos(2.0 * np.pi * y))
    bumps = 350.0 * (np.sin(2.0 * np.pi * 5.0 * x) * np.sin(np.pi * 7.0 * y)) ** 2
    return np.maximum(4000.0 - slope - ridge - bumps, 250.0)


def find_levels(el, depth, zbar):
    """bottom level index of every element from the node depths, as the reference's partitioner derives it (find_levels,
    src/fvom_init.F90:657-870): the first level whose mid-depth lies below the mean depth of the three nodes, at least level 5, then
    the layer-by-layer removal of isolated cells (a wet cell needs two wet neighbours; the test `elems(i) > 1` of the reference, which
    never counts element 1 as a neighbour, is kept).  el: (E,3) 0-based nodes; depth, zbar negative down.  Returns (elvls, nlvls), 1-based."""
    E, nl = el.shape[0], zbar.size
    Z = 0.5 * (zbar[:-1] + zbar[1:])
    dmean = depth[el].sum(1) / 3.0
    below = Z[None, :] < dmean[:, None]
    lev = np.where(below.any(1), below.argmax(1) + 1, nl)
    lev = np.where(dmean >= 0, THERS_ZBAR_LEV, lev)
    lev = np.maximum(lev, THERS_ZBAR_LEV)
    # element neighbours across the three edges (0 = none), 1-based
    pairs = np.concatenate([el[:, [0, 1]], el[:, [1, 2]], el[:, [2, 0]]])
    key = pairs.min(1).astype(np.int64) * (depth.size + 1) + pairs.max(1)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same = ks[1:] == ks[:-1]
    nb = np.zeros(3 * E, dtype=np.int64)
    a, b = order[:-1][same], order[1:][same]
    nb[a] = b % E + 1; nb[b] = a % E + 1
    nb = nb.reshape(3, E).T
    for nz in range(THERS_ZBAR_LEV + 1, nl + 1):
        for _ in range(1000):
            wet = lev >= nz
            nbwet = np.where(nb > 1, lev[np.maximum(nb, 1) - 1] >= nz, False)
            iso = wet & (nbwet.sum(1) < 2)
            if not iso.any():
                break
            # (the reference sweeps the elements in order and lets a change act at once; on the smooth analytic bathymetry of this package
            # the isolated cells of a layer do not touch each other, so the simultaneous update gives the same result -- asserted)
            touched = np.zeros(E, dtype=bool)
            touched[np.maximum(nb[iso], 1).ravel() - 1] = True
            assert not (touched & iso).any(), "find_levels: neighbouring isolated cells -- sequential sweep needed"
            assert nz - 1 >= THERS_ZBAR_LEV
            lev = np.where(iso, nz - 1, lev)
    nlv = np.zeros(depth.size, dtype=np.int64)
    np.maximum.at(nlv, el.ravel(), np.repeat(lev, 3))
    return lev, nlv


def build_basin(base_dir, outdir, levels, layers=47):
    """the `basin` variant: the channel refined `levels` times, `layers` stretched layers, bottom depths from basin_depth, element and
    node levels by the reference partitioner's rule.  Returns (nodes, elements, nl)."""
    N, E, nl = build(base_dir, outdir, levels, layers)
    t = open(os.path.join(outdir, "nod2d.out")).read().split()
    a = np.array(t[1:1 + 4 * N], dtype=np.float64).reshape(N, 4)
    t = open(os.path.join(outdir, "elem2d.out")).read().split()
    el = np.array(t[1:1 + 3 * E], dtype=np.int64).reshape(E, 3) - 1
    t = open(os.path.join(outdir, "aux3d.out")).read().split()
    zb = np.array(t[1:1 + nl], dtype=np.float64)
    depth = np.round(basin_depth(a[:, 1], a[:, 2]), 3)           # (what the %.3f of aux3d.out keeps)
    zbar = -np.abs(zb)
    dn = -depth
    dn = np.minimum(dn, zbar[THERS_ZBAR_LEV - 1])               # depth thresholding of find_levels (fvom_init.F90:695)
    elv, nlv = find_levels(el, dn, zbar)
    with open(os.path.join(outdir, "aux3d.out"), "w") as f:
        f.write(f"{nl}\n" + "\n".join(f"{z:.10g}" for z in np.abs(zb)) + "\n")
        np.savetxt(f, depth, fmt="%.3f")
    np.savetxt(os.path.join(outdir, "elvls.out"), elv, fmt="%d")
    np.savetxt(os.path.join(outdir, "nlvls.out"), nlv, fmt="%d")
    return N, E, nl


def dt_for(levels):
    return BASE_DT / 2 ** levels


def mesh_kw(levels):
    """keyword arguments of Mesh.load for this workload (the reference's namelist values of test_souf)"""
    return dict(force_rotation=False, cyclic_length_deg=CYCLIC_DEG, dt=dt_for(levels), K_hor=10.0)


def basin_param_kw(levels):
    """the reference's default physics (config/namelist.oce: KPP, GM, Redi, JM EOS) on the basin variant, no toy hooks"""
    return dict(dt=dt_for(levels), state_equation=1, mix_scheme="KPP", Fer_GM=True, Redi=True, with_diffusion=True, toy_soufflet=False,
                K_hor=3000.0, cyclic_length_deg=CYCLIC_DEG)


def basin_mesh_kw(levels):
    return dict(force_rotation=False, cyclic_length_deg=CYCLIC_DEG, dt=dt_for(levels), K_hor=3000.0)


def param_kw(levels):
    return dict(dt=dt_for(levels), state_equation=0, mix_scheme="PP", with_diffusion=True, toy_soufflet=True, K_hor=10.0,
                cyclic_length_deg=CYCLIC_DEG)
