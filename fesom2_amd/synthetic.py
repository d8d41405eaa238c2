"""Synthetic initial tracer fields for meshes that have no climatology file.

The reference initialises T/S from NetCDF climatologies (gen_ic3d.F90), which are
not available offline.  This analytic replacement uses only +,-,*,/ so that it is
bit-reproducible on every host (no libm): both the reference oracle (through its
`do_ic3d` stub, oracle/ref/stubs.F90) and this package start from these bits.

    x   = lat_deg/90,  c2 = (1-x^2)^2            (~cos^2(lat))
    e_H = 1/(1 - z/H)^2                          (z<=0 mid-level depth, decays with depth)
    T   = 1 + 24*e_700*c2 + 0.5*e_500*w(lon)     w = x_l*(1-x_l^2)*2.598.., x_l = lon_deg/180
    S   = 34.7 - 0.8*e_400*(2*c2-1)
"""
import os
import numpy as np


def read_nod2d(meshdir):
    with open(os.path.join(meshdir, "nod2d.out")) as f:
        n = int(f.readline().split()[0])
        a = np.loadtxt(f, dtype=np.float64, max_rows=n)
    return a[:, 1].copy(), a[:, 2].copy()   # lon, lat in degrees


def read_zbar(meshdir):
    with open(os.path.join(meshdir, "aux3d.out")) as f:
        toks = f.read().split()
    nl = int(toks[0])
    zbar = np.array([float(t) for t in toks[1:1 + nl]], dtype=np.float64)
    if zbar[1] > 0:
        zbar = -zbar
    return nl, zbar


def analytic_ts(meshdir, contrast=1.0):
    """Returns T, S as float64 arrays of shape (nod2D, nl-1) (level index fastest in memory).  `contrast` scales the horizontal
    variations around the profile of 30 degrees latitude (1.0 = the fields of the golden runs; small values give a nearly
    balanced, weakly forced ocean for the large synthetic meshes, whose untuned set-up is unstable under the full contrast)."""
    lon, lat = read_nod2d(meshdir)
    nl, zbar = read_zbar(meshdir)
    Z = 0.5 * (zbar[:-1] + zbar[1:])
    x = lat / 90.0
    c2 = (1.0 - x * x) * (1.0 - x * x)
    lon = np.where(lon > 180.0, lon - 360.0, lon)
    xl = lon / 180.0
    w = xl * (1.0 - xl * xl) * 2.598
    def e(H):
        q = 1.0 - Z / H
        return 1.0 / (q * q)
    if contrast != 1.0:
        c0 = (1.0 - 1.0 / 9.0) ** 2
        c2 = c0 + contrast * (c2 - c0)
        w = contrast * w
    T = 1.0 + 24.0 * e(700.0)[None, :] * c2[:, None] + 0.5 * e(500.0)[None, :] * w[:, None]
    S = 34.7 - 0.8 * e(400.0)[None, :] * (2.0 * c2[:, None] - 1.0)
    return np.ascontiguousarray(T), np.ascontiguousarray(S)


def write_ic_files(meshdir, outdir):
    T, S = analytic_ts(meshdir)
    T.tofile(os.path.join(outdir, "ic_T.bin"))
    S.tofile(os.path.join(outdir, "ic_S.bin"))
    return T, S


def analytic_forcing(mesh):
    """Analytic surface forcing on a (global or rank-local) mesh: nodal wind stress, heat flux (W/m2, positive up) and fresh-water
    flux (m/s) with both signs of the surface buoyancy flux, and the element wind stress as the mean of the nodal one
    (ice_oce_coupling.F90:62-68 with a_ice = 0).  Same formulas as the reference harness uses for the golden runs."""
    lon, lat = mesh.geo_coord_nod2D[:, 0], mesh.geo_coord_nod2D[:, 1]
    f = {"stress_atmoce_x": 0.1 * np.cos(3.0 * lat), "stress_atmoce_y": 0.03 * np.sin(2.0 * lon),
         "heat_flux": 150.0 * np.sin(2.0 * lon + 1.0) * np.cos(lat), "water_flux": 2.0e-8 * np.cos(3.0 * lon)}
    en = mesh.elem2D_nodes[:mesh.myDim_elem2D] - 1
    f["stress_surf"] = np.stack([f["stress_atmoce_x"][en].sum(1) / 3.0, f["stress_atmoce_y"][en].sum(1) / 3.0], axis=1)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in f.items()}


def analytic_ice(mesh):
    """Analytic ice state for mo_length of mo_convect (use_momix): ice-free, partly and fully covered regions, ice drift.  Same formulas as the
    reference harness (oracle/ref/driver.F90)."""
    lon, lat = mesh.geo_coord_nod2D[:, 0], mesh.geo_coord_nod2D[:, 1]
    f = {"a_ice": np.minimum(1.0, np.maximum(0.0, -0.9 - 1.6 * np.sin(lat) + 0.25 * np.cos(3.0 * lon))),
         "u_ice": 0.08 * np.sin(lon) * np.cos(lat), "v_ice": 0.05 * np.cos(2.0 * lon)}
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in f.items()}


def analytic_surface_potentials(mesh):
    """Analytic ice / snow thickness, air pressure and tidal potential for the surface pressure gradient of compute_vel_rhs (use_floatice, l_mslp,
    use_global_tides).  Same formulas as the reference harness (oracle/ref/driver.F90)."""
    lon, lat = mesh.geo_coord_nod2D[:, 0], mesh.geo_coord_nod2D[:, 1]
    mi = np.maximum(0.0, 12.0 * (np.sin(lat) * np.sin(lat) - 0.55)) * (1.0 + 0.5 * np.cos(2.0 * lon))
    f = {"m_ice": mi, "m_snow": 0.2 * mi, "press_air": 101325.0 + 1500.0 * np.sin(2.0 * lon + 0.5) * np.cos(lat), "ssh_gp": 2.5 * np.sin(2.0 * lon) * np.cos(lat) * np.cos(lat)}
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in f.items()}


def analytic_sw_3d(mesh, heat_flux):
    """Penetrating short-wave flux / vcpw [K m/s] (nl, N) for use_sw_pene: half of the positive part of `heat_flux`, decaying
    over ~15 m; +,-,*,/ only, the same operations as the reference harness (bit-identical values)."""
    q = 1.0 - mesh.zbar / 15.0
    sw = (np.maximum(heat_flux, 0.0) / 4.2e6 * 0.5)[:, None] / (q * q)[None, :]
    lev = np.arange(1, mesh.nl + 1)[None, :]
    return np.ascontiguousarray(np.where(lev <= mesh.nlevels_nod2D[:, None], sw, 0.0))
