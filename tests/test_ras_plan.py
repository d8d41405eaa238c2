"""RAS-Chebyshev SSH preconditioner (csrc/ras_host.h): the plan the product's host code builds (patches of the row graph, overlap
rings, fp32 patch operators, Chebyshev coefficients) equals the CPU checker's own restatement entry for entry, and the checker's
preconditioned BiCGstab converges in a fraction of the Jacobi iterations.  CPU only (no compute call into the HIP library)."""
import ctypes as C
import os
import numpy as np
import pytest

from fesom2_amd import _lib, workloads
from oracle_lib import Oracle, ORC_LIB, build as build_oracle

PI = C.POINTER(C.c_int)


def _csr(mesh):
    n = mesh.myDim_nod2D
    rp = (np.array(mesh.ssh_rowptr[: n + 1]) - mesh.ssh_rowptr[0]).astype(np.int32)
    ci = (np.array(mesh.ssh_colind_loc) - 1).astype(np.int32)
    return n, rp, ci, np.array(mesh.ssh_values, dtype=np.float64)


def _export(fn, n, rp, ci, vals):
    dims = np.zeros(6, np.int32)
    args = (C.c_int(n), rp.ctypes.data_as(PI), ci.ctypes.data_as(PI), vals.ctypes.data_as(C.POINTER(C.c_double)), dims.ctypes.data_as(PI))
    assert fn(*args, None, None, None, None, None, None, None) == 0
    P, NS, rpt, woff, deg, next_ = (int(v) for v in dims)
    out = dict(perm=np.zeros(n, np.int32), pinfo=np.zeros(4 * P, np.int32), extq=np.zeros(next_, np.int32), lv=np.zeros(P * woff * NS, np.float32),
               lc=np.zeros(P * woff * NS, np.uint16), dsc=np.zeros(P * NS, np.float64), cheb=np.zeros(128, np.float64))
    assert fn(*args, *(out[k].ctypes.data_as(C.c_void_p) for k in ("perm", "pinfo", "extq", "lv", "lc", "dsc", "cheb"))) == 0
    out["dims"] = dims
    return out


@pytest.mark.parametrize("which", ["pi", "channel1"])
def test_product_plan_equals_checker_plan(which):
    wl = workloads.pi("pp") if which == "pi" else workloads.channel(1)
    mesh = wl.load_mesh()
    n, rp, ci, vals = _csr(mesh)
    if not os.path.exists(ORC_LIB):
        build_oracle()
    orc = C.CDLL(ORC_LIB)
    lib = _lib.load()
    a = _export(lib.fesom_ras_plan_export, n, rp, ci, vals)
    b = _export(orc.orc_ras_plan_export, n, rp, ci, vals)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    P, NS, rpt, woff, deg, next_ = (int(v) for v in a["dims"])
    pinfo = a["pinfo"].reshape(P, 4)
    # every row is owned by exactly one patch; the owned rows of a patch are a run of consecutive positions that heads its row list
    assert sorted(a["perm"].tolist()) == list(range(n))
    assert pinfo[:, 1].sum() == n and (pinfo[:, 1] <= 768).all() and (pinfo[:, 3] <= NS).all() and (pinfo[:, 3] >= pinfo[:, 1]).all()
    for p in range(P):
        own0, no, eoff, ne = pinfo[p]
        assert np.array_equal(a["extq"][eoff:eoff + no], np.arange(own0, own0 + no))
        assert len(set(a["extq"][eoff:eoff + ne].tolist())) == ne
    assert a["lc"].max() < NS and 1.0 < a["cheb"][127] <= 2.0 + 1e-12


def test_checker_ras_bicgstab_converges_fast_on_the_channel():
    """channel refined once (11 450 rows): the checker's RAS-preconditioned solve needs far fewer iterations than Jacobi and returns the same d_eta"""
    wl = workloads.channel(1)
    mesh = wl.load_mesh()
    res = {}
    for pc in (0, 1):
        orc = Oracle(mesh, wl.params(solver_precond=pc))
        st, aux, _ = wl.initial_state(mesh)
        orc.set_state(st)
        for k, v in aux.items():
            orc.set(k, v)
        orc.call("compute_zonal_mean_ini"); orc.call("compute_zonal_mean")
        its = []
        for s in (1, 2, 3):
            orc.call("step", s)
            its.append(orc.solver_iterations)
        res[pc] = (its, orc.get("d_eta").copy(), orc.get("eta_n").copy())
    assert max(res[1][0]) <= 25 and min(res[0][0]) >= 2 * max(res[1][0]), (res[0][0], res[1][0])
    # both solves stop at ||scaled residual|| < 1e-10 (the reference's rule): the solutions agree to that tolerance times the conditioning of A_s
    print("iterations jacobi / ras:", res[0][0], res[1][0], "max |d d_eta|", np.abs(res[0][1] - res[1][1]).max())
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-8 and np.abs(res[0][2] - res[1][2]).max() < 1e-8
