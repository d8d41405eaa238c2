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


def dt_for(levels):
    return BASE_DT / 2 ** levels


def mesh_kw(levels):
    """keyword arguments of Mesh.load for this workload (the reference's namelist values of test_souf)"""
    return dict(force_rotation=False, cyclic_length_deg=CYCLIC_DEG, dt=dt_for(levels), K_hor=10.0)


def param_kw(levels):
    return dict(dt=dt_for(levels), state_equation=0, mix_scheme="PP", with_diffusion=True, toy_soufflet=True, K_hor=10.0,
                cyclic_length_deg=CYCLIC_DEG)
