"""Host side of the reference's Soufflet channel toy set-up (`toy_ocean=.true., which_toy='soufflet'`): the analytic
initial state of `initial_state_soufflet` (src/toy_channel_soufflet.F90:220-343) -- temperature of the baroclinic jet
plus perturbation, constant salinity 35, geostrophically balanced zonal flow, the redefined Coriolis parameter and
the relaxation targets Tclim / Uclim.  It is the caller's job in the reference too (ocean_setup, untimed); the hooks
that sit on the step path (zonal means, velocity / temperature relaxation) are device kernels behind
`fesom_gpu_step`.

Arithmetic follows the Fortran expressions operation by operation (left-to-right, `**2` = x*x); the transcendental
calls go through Python's `math` (the C library, like the reference's build), so the result is bit-identical to the
reference's own initial state on the soufflet test mesh (`tests/test_soufflet.py` checks it against the digests).
"""
import math
import numpy as np

PI = 3.14159265358979          # o_PARAM pi (src/oce_modules.F90:11), NOT math.pi
R_EARTH = 6367500.0
DENSITY_0 = 1030.0
G = 9.81
# module constants of Toy_Channel_Soufflet (src/toy_channel_soufflet.F90:18-39)
LAT0, YSIZE, XSIZE, LJET = 0.0, 2000000.0, 90018410.49779853, 1600000.0
RHOMAX, SB, ZSIZE = 27.75, 9.8e-6, 4000.0
DRHO_NO, DRHO_SO, Z_NO, Z_SO, DZ_NO, DZ_SO = 1.41, 1.4, -400.0, -1000.0, 300.0, 700.0
DRHOSURF_NO, DRHOSURF_SO, ZSURF = 0.0, 1.5, -300.0
FORC_UPDATE = 10               # forcing-update period of the toy module: zonal means every 10 steps


def _profile(Z, z0, dz0, drho, drhosurf):
    """north/south density profile -> temperature profile (:243-261)"""
    out = np.zeros(len(Z))
    for k, z in enumerate(Z):
        zz = float(z)
        q = ((zz - z0) + abs(zz - z0)) / 1.3 / dz0
        d = z0 + (zz - z0) * math.sqrt(1 + 0.5 * (q * q))
        rho = RHOMAX - SB * (zz + ZSIZE) - 0.5 * drho * (1 + math.tanh((d - z0) / dz0)) - \
            1.0 / (2 * math.tanh(1.0)) * drhosurf * (1 + math.tanh((ZSURF - zz) / ZSURF))
        out[k] = 10.0 - (rho - RHOMAX) / (0.00025 * DENSITY_0)
    return out


def initial_state(mesh, st):
    """Fill `st` (fesom2_amd.mesh.State of `mesh.initial_state(2)`) with the Soufflet initial state; overwrite
    `mesh.coriolis` in place.  Returns dict(Tclim=(N, nl-1), Uclim=(E, nl-1))."""
    nlm1 = mesh.nl - 1
    Z, zbar = np.array(mesh.Z, dtype=np.float64), np.array(mesh.zbar, dtype=np.float64)
    N = mesh.myDim_nod2D + mesh.eDim_nod2D
    myE = mesh.myDim_elem2D
    t_no = _profile(Z, Z_NO, DZ_NO, DRHO_NO, DRHOSURF_NO)
    t_so = _profile(Z, Z_SO, DZ_SO, DRHO_SO, DRHOSURF_SO)
    lon, lat = mesh.coord_nod2D[:, 0], mesh.coord_nod2D[:, 1]
    nlev_n = mesh.nlevels_nod2D
    T = np.zeros((N, nlm1))
    st.tr_arr[1][...] = 35.0
    # 2-D profile (:266-281)
    for n in range(N):
        dst = (float(lat[n]) - LAT0) * R_EARTH
        yn = PI * (YSIZE / LJET) * (dst / YSIZE - 0.5) + PI / 2.0
        if yn < 0:
            Fy = 1.0
        elif yn > PI:
            Fy = 0.0
        else:
            Fy = 1.0 - (yn - math.sin(yn) * math.cos(yn)) / PI
        k = int(nlev_n[n]) - 1
        T[n, :k] = t_so[:k] + (t_no[:k] - t_so[:k]) * (1.0 - Fy)
    Tclim = T.copy()
    # perturbation (:290-297)
    ez = np.array([math.exp(2 * float(z) / ZSIZE) for z in Z])
    for n in range(N):
        dst = (float(lat[n]) - LAT0) * R_EARTH
        x = float(lon[n])
        a = 0.1 * math.sin(2 * PI * dst / YSIZE)
        b = math.sin(8 * PI * x * R_EARTH / XSIZE) + 0.5 * math.sin(3 * PI * x * R_EARTH / XSIZE)
        k = int(nlev_n[n]) - 1
        T[n, :k] = T[n, :k] - a * ez[:k] * b
    st.tr_arr[0][...] = T
    st.tr_arr_old[...] = st.tr_arr
    # Coriolis of the Soufflet paper (:303-307) and the thermal-wind balanced zonal flow (:309-322)
    en = mesh.elem2D_nodes[:myE] - 1
    gs = mesh.gradient_sca
    nlev = mesh.nlevels
    cor = mesh.coriolis
    UV = st.UV
    UV[...] = 0.0
    for e in range(myE):
        n1, n2, n3 = (int(v) for v in en[e])
        dst = (((float(lat[n1]) + float(lat[n2])) + float(lat[n3])) / 3.0 - LAT0) * R_EARTH - YSIZE / 2
        cor[e] = 1.0e-4 + dst * 1.6e-11
        fac = (-(0.00025 * DENSITY_0) * G / DENSITY_0 / float(cor[e]))
        k = int(nlev[e]) - 1
        dN = fac * ((gs[e, 3] * Tclim[n1, :k] + gs[e, 4] * Tclim[n2, :k]) + gs[e, 5] * Tclim[n3, :k])
        u = np.zeros(k)
        u[k - 1] = dN[k - 1] * (Z[k - 1] - zbar[k])
        for nz in range(k - 2, -1, -1):          # 0-based level nz <-> Fortran nz+1
            u[nz] = (u[nz + 1] + dN[nz + 1] * (zbar[nz + 1] - Z[nz + 1])) + dN[nz] * (Z[nz] - zbar[nz + 1])
        UV[e, :k, 0] = u
    Uclim = np.ascontiguousarray(UV[:, :, 0]).copy()
    return dict(Tclim=np.ascontiguousarray(Tclim), Uclim=Uclim)


def fcheck_means(sumT, sumS, sumU, sumV, nsteps):
    """The reference CI's check values (setups/test_souf/setup.yml:82-88): unweighted mean over all (level, entity)
    entries of the 1-day time mean.  sum* = running sums over the steps of tr_arr(:,:,1), tr_arr(:,:,2), UV(1), UV(2)."""
    mT, mS, mU, mV = sumT / nsteps, sumS / nsteps, sumU / nsteps, sumV / nsteps
    return dict(temp=float(mT.mean()), salt=float(mS.mean()), sst=float(mT[:, 0].mean()), u=float(mU.mean()), v=float(mV.mean()))
