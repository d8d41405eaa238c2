#!/usr/bin/env python3
"""CPU prototype for the SSH preconditioner study (numpy/scipy, test infrastructure -- uses the oracle to produce real
(operator, rhs, initial guess) triples): BiCGstab on the row-scaled operator with the stop rule of the reference
(||r||^2 < 1e-20 on the scaled residual), preconditioners: Jacobi, Chebyshev polynomial, multicolour SGS, aggregation AMG.
usage: solver_proto.py pi|chanL [nsteps]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from fesom2_amd import workloads


def capture(wl, nsteps, every=1):
    from oracle_lib import Oracle
    from parity_chain import full_chain
    mesh = wl.load_mesh()
    par = wl.params(solver_x0_order=0)
    orc = Oracle(mesh, par)
    st, aux, forcing = wl.initial_state(mesh)
    orc.set_state(st)
    for k, v in aux.items():
        orc.set(k, v)
    for k, v in (forcing or {}).items():
        orc.set(k, v)
    if wl.name == "channel":
        orc.call("compute_zonal_mean_ini"); orc.call("compute_zonal_mean")
    n = mesh.myDim_nod2D
    rp = np.array(mesh.ssh_rowptr[: n + 1]) - mesh.ssh_rowptr[0]
    ci = np.array(mesh.ssh_colind_loc) - 1
    out = []
    gm, redi, kpp = bool(par.Fer_GM), bool(par.Redi), par.mix_scheme == 1
    for s in range(1, nsteps + 1):
        if wl.name == "channel" and s % 10 == 0:
            orc.call("compute_zonal_mean")
        for routine, arg, _ in full_chain(2, gm=gm, redi=redi, kpp=kpp):
            if routine == "solve_ssh" and (s % every == 0 or s <= 3):
                vals = orc.get("ssh_values").copy()
                A = sp.csr_matrix((vals, ci.copy(), rp.copy()), shape=(n, n))
                out.append((s, A, orc.get("ssh_rhs")[:n].copy(), orc.get("d_eta")[:n].copy()))
            orc.call(routine, arg)
            if wl.name == "channel" and routine == "solve_ssh":
                orc.call("relax_zonal_vel")
            if wl.name == "channel" and routine == "diff_tracers_ale":
                orc.call("relax_zonal_temp")
        if s % every == 0 or s <= 3:
            print("captured step", s, "oracle its", orc.solver_iterations, flush=True)
    return out


def bicgstab(Aop, b, x0, M=None, tol2=1e-20, maxit=2000):
    """right-preconditioned BiCGstab, same recurrences as csrc/solver.hip (y = preconditioned unknown)"""
    M = M or (lambda v: v)
    x = x0.copy()
    r = b - Aop(x)
    r0 = r.copy()
    rr = r @ r
    rho = alpha = omega = 1.0
    v = np.zeros_like(b); p = np.zeros_like(b)
    it = 0
    rho_new = rr
    while rr >= tol2 and it < maxit:
        beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v)
        ph = M(p)
        v = Aop(ph)
        alpha = rho_new / (r0 @ v)
        s = r - alpha * v
        sh = M(s)
        t = Aop(sh)
        tt = t @ t
        omega = (t @ s) / tt if tt > 0 else 0.0
        x = x + alpha * ph + omega * sh
        r = s - omega * t
        rho, rho_new = rho_new, -omega * (r0 @ t)
        rr = r @ r
        it += 1
    return x, it, np.sqrt(rr)


def scaled(A):
    sc = 1.0 / np.asarray(abs(A).sum(axis=1)).ravel()
    return sp.diags(sc) @ A, sc


def cheb_prec(As, deg, lmin_frac=None):
    """Chebyshev polynomial in D^-1 A_s of degree `deg` approximating the inverse on [lmin, lmax]"""
    d = As.diagonal()
    Dinv = 1.0 / d
    B = sp.diags(Dinv) @ As
    lmax = spla.eigs(B, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0].real * 1.02
    lmin = lmax / (lmin_frac or 30.0)
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)

    def M(r):
        # standard Chebyshev iteration for B z = Dinv r, z0 = 0
        rb = Dinv * r
        sigma = theta / delta
        rho = 1.0 / sigma
        dd = rb / theta
        z = dd.copy()
        for _ in range(deg - 1):
            res = rb - B @ z
            rho_n = 1.0 / (2.0 * sigma - rho)
            dd = rho_n * rho * dd + 2.0 * rho_n / delta * res
            z = z + dd
            rho = rho_n
        return z
    return M, lmax


def colour(A):
    n = A.shape[0]
    indptr, indices = A.indptr, A.indices
    col = -np.ones(n, dtype=np.int64)
    for i in range(n):
        used = set(col[indices[indptr[i]:indptr[i + 1]]])
        c = 0
        while c in used:
            c += 1
        col[i] = c
    return col


def mc_sgs_prec(As):
    """symmetric Gauss-Seidel in multicolour order (= ILU(0)-like, parallel per colour)"""
    col = colour(sp.csr_matrix(As))
    order = np.argsort(col, kind="stable")
    P = sp.csr_matrix((np.ones(len(order)), (np.arange(len(order)), order)))
    Ap = (P @ As @ P.T).tocsr()
    L = sp.tril(Ap, 0).tocsr(); U = sp.triu(Ap, 0).tocsr(); D = Ap.diagonal()

    def M(r):
        rp_ = P @ r
        y = spla.spsolve_triangular(L, rp_, lower=True)
        z = spla.spsolve_triangular(U, D * y, lower=False)
        return P.T @ z
    return M, int(col.max() + 1)


def ilu_prec(As, fill=1):
    ilu = spla.spilu(sp.csc_matrix(As), drop_tol=0.0, fill_factor=fill, permc_spec="NATURAL", diag_pivot_thresh=0.0)
    return lambda r: ilu.solve(r)


def aggregate(A, theta=0.0):
    """greedy aggregation (root + its unaggregated strong neighbours; leftovers join a neighbouring aggregate)"""
    n = A.shape[0]
    indptr, indices = A.indptr, A.indices
    agg = -np.ones(n, dtype=np.int64)
    na = 0
    for i in range(n):
        nb = indices[indptr[i]:indptr[i + 1]]
        if agg[i] < 0 and np.all(agg[nb] < 0):
            agg[nb] = na; agg[i] = na; na += 1
    for i in range(n):
        if agg[i] < 0:
            nb = indices[indptr[i]:indptr[i + 1]]
            cand = agg[nb][agg[nb] >= 0]
            if len(cand):
                agg[i] = cand[0]
            else:
                agg[i] = na; na += 1
    return agg, na


class AMG:
    def __init__(self, As, coarse=64, nu=1, omega=0.7, smoother="jacobi", kcycle=False, over=1.0, maxlev=10, sa=0.0, cheb=0):
        self.lv = []
        A = sp.csr_matrix(As)
        while True:
            d = A.diagonal()
            lev = dict(A=A, Dinv=1.0 / d)
            if smoother == "l1":
                lev["Dinv"] = 1.0 / np.asarray(abs(A).sum(axis=1)).ravel() * 1.0
            self.lv.append(lev)
            if A.shape[0] <= coarse or len(self.lv) >= maxlev:
                break
            agg, na = aggregate(A)
            P = sp.csr_matrix((np.ones(A.shape[0]), (np.arange(A.shape[0]), agg)), shape=(A.shape[0], na))
            if sa > 0.0:                      # smoothed aggregation: P = (I - sa * D^-1 A) P_tent
                P = (P - sa * (sp.diags(1.0 / d) @ A @ P)).tocsr()
            lev["P"] = P
            A = (P.T @ A @ P).tocsr()
        self.nu, self.omega, self.over = nu, omega, over
        last = self.lv[-1]["A"]
        self.coarse_lu = spla.splu(sp.csc_matrix(last))
        self.sizes = [l["A"].shape[0] for l in self.lv]
        self.nnz = [round(l["A"].nnz / l["A"].shape[0], 1) for l in self.lv]
        self.cheb = cheb
        if cheb:
            for l in self.lv[:-1]:
                B = sp.diags(l["Dinv"]) @ l["A"]
                l["lmax"] = abs(spla.eigs(B, k=1, which="LM", return_eigenvectors=False, tol=1e-2)[0]) * 1.05

    def smooth(self, lev, z, r):
        A, Dinv = lev["A"], lev["Dinv"]
        if not self.cheb:
            for _ in range(self.nu):
                z = z + self.omega * Dinv * (r - A @ z)
            return z
        lmax = lev["lmax"]; lmin = lmax / 4.0
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta; rho = 1.0 / sigma
        res = Dinv * (r - A @ z)
        dd = res / theta
        z = z + dd
        for _ in range(self.cheb - 1):
            res = Dinv * (r - A @ z)
            rho_n = 1.0 / (2.0 * sigma - rho)
            dd = rho_n * rho * dd + 2.0 * rho_n / delta * res
            z = z + dd
            rho = rho_n
        return z

    def cycle(self, k, r):
        lev = self.lv[k]
        if k == len(self.lv) - 1:
            return self.coarse_lu.solve(r)
        A, Dinv, P = lev["A"], lev["Dinv"], lev["P"]
        z = self.smooth(lev, np.zeros_like(r), r)
        rc = P.T @ (r - A @ z)
        zc = self.cycle(k + 1, rc)
        z = z + self.over * (P @ zc)
        return self.smooth(lev, z, r)

    def __call__(self, r):
        return self.cycle(0, r)


def main():
    what = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    wl = workloads.pi(sys.argv[3] if len(sys.argv) > 3 else "pp") if what == "pi" else workloads.channel(int(what[4:]))
    caps = capture(wl, nsteps, every=max(1, nsteps // 4))
    A0s, _ = scaled(caps[0][1])
    t0 = time.time()
    precs = {"jacobi": None}
    d0 = A0s.diagonal()
    precs["jacobi"] = lambda r, d=d0: r / d
    if "--amg-only" not in sys.argv:
        for deg in (2, 3, 4):
            precs[f"cheb{deg}"] = cheb_prec(A0s, deg)[0]
        M, nc = mc_sgs_prec(A0s); precs[f"mcSGS({nc} colours)"] = M
        precs["ilu0-natural"] = ilu_prec(A0s, 1)
    for kw in (dict(nu=1, over=1.5), dict(nu=1, over=1.0, sa=0.66), dict(nu=1, over=1.0, sa=0.5), dict(nu=2, over=1.0, sa=0.66), dict(cheb=2, over=1.5), dict(cheb=2, sa=0.66), dict(cheb=3, sa=0.66)):
        amg = AMG(A0s, **kw)
        precs[f"amg {kw} {amg.sizes} nnz/row {amg.nnz}"] = amg
    print("setup s", round(time.time() - t0, 1))
    for name, M in precs.items():
        res = []
        for s, A, b, x0 in caps:
            As, sc = scaled(A)
            x, it, rn = bicgstab(lambda v: As @ v, b * sc, x0, M)
            res.append(it)
        print(f"{name:50s} iterations per captured step {res}", flush=True)


if __name__ == "__main__":
    main()
