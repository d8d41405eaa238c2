"""GPU parity on the Soufflet channel = the reference's CI known-answer case (setups/test_souf/setup.yml): cyclic channel,
zstar + partial cells, linear EOS, PP mixing, toy relaxation hooks.  (i) HIP == oracle bit for bit, routine by routine over
3 steps including the toy hooks; (ii) 72 steps through fesom_gpu_run_steps reproduce the CI's `fcheck` numbers and the
oracle's final state bit for bit."""
import numpy as np
import pytest

from parity_chain import full_chain, compare
from test_soufflet import soufflet_setup, check_fcheck

pytestmark = pytest.mark.gpu


def toy_chain():
    ch = []
    for routine, arg, fields in full_chain(2):
        ch.append((routine, arg, fields))
        if routine == "solve_ssh":
            ch.append(("relax_zonal_vel", 0, ["UV_rhs"]))
        if routine == "diff_tracers_ale":
            ch.append(("relax_zonal_temp", 0, ["tr_arr"]))
    return ch


@pytest.fixture(scope="module")
def setup(built):
    from fesom2_amd.core import OceanCore
    from oracle_lib import Oracle
    mesh, par, st, aux = soufflet_setup()
    gpu = OceanCore(mesh, par)
    orc = Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    for k in ("Tclim", "Uclim"):
        gpu.set(k, aux[k]); orc.set(k, aux[k])
    orc.call("compute_zonal_mean_ini")                      # (the library prepares these tables in fesom_gpu_init)
    gpu.call("compute_zonal_mean"); orc.call("compute_zonal_mean")
    yield mesh, par, gpu, orc, st, aux
    gpu.close()


def test_soufflet_chain_bitwise(setup):
    mesh, par, gpu, orc, st, aux = setup
    for f in ("toy_zvel", "toy_ztem"):
        ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
        assert ok, msg
    failures = []
    for step in range(1, 4):
        for routine, arg, fields in toy_chain():
            gpu.call(routine, arg)
            orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
            if failures:
                break
        if failures:
            break
    assert not failures, "\n".join(failures)
    assert gpu.solver_iterations == orc.solver_iterations


def test_soufflet_linfs_chain_bitwise(built):
    """linear free surface + full cells on the same channel (pinned against the reference in tests/test_soufflet.py)"""
    from fesom2_amd.core import OceanCore
    from oracle_lib import Oracle
    mesh, par, st, aux = soufflet_setup("linfs", False)
    gpu = OceanCore(mesh, par)
    orc = Oracle(mesh, par)
    gpu.upload_state(st); orc.set_state(st)
    for k in ("Tclim", "Uclim"):
        gpu.set(k, aux[k]); orc.set(k, aux[k])
    orc.call("compute_zonal_mean_ini")
    gpu.call("compute_zonal_mean"); orc.call("compute_zonal_mean")
    failures = []
    for step in range(1, 3):
        for routine, arg, fields in toy_chain():
            if routine == "update_stiff_mat_ale":
                continue
            gpu.call(routine, arg); orc.call(routine, arg)
            for f in fields:
                ok, msg = compare(f, gpu.get(f, orc.count(f)), orc.get(f))
                if not ok:
                    failures.append(f"step {step} {routine}({arg}) {msg}")
        if failures:
            break
    gpu.close()
    assert not failures, "\n".join(failures[:10])


def test_soufflet_known_answer(setup):
    """fresh start, 72 steps (1 day): time means of T, S, u, v == the reference CI's fcheck values; HIP == oracle bitwise"""
    from fesom2_amd import toy_soufflet
    from fesom2_amd.core import OceanCore
    from oracle_lib import Oracle
    mesh, par, gpu, orc, st, aux = setup
    gpu.close()
    gpu = OceanCore(mesh, par)
    orc2 = Oracle(mesh, par)
    gpu.upload_state(st); orc2.set_state(st)
    for k in ("Tclim", "Uclim"):
        gpu.set(k, aux[k]); orc2.set(k, aux[k])
    orc2.call("compute_zonal_mean_ini")
    gpu.call("compute_zonal_mean"); orc2.call("compute_zonal_mean")
    nlm1 = mesh.nl - 1
    sT = np.zeros((mesh.nod2D, nlm1)); sS = np.zeros_like(sT); sU = np.zeros((mesh.elem2D, nlm1)); sV = np.zeros_like(sU)
    for n in range(1, 73):
        gpu.run_steps(n, 1)
        tr = gpu.get("tr_arr", 2 * mesh.nod2D * nlm1).reshape(2, -1, nlm1); uv = gpu.get("UV", 2 * mesh.elem2D * nlm1).reshape(-1, nlm1, 2)
        sT += tr[0]; sS += tr[1]; sU += uv[:, :, 0]; sV += uv[:, :, 1]
        if n <= 12:                                          # covers the first zonal-mean refresh at step 10
            orc2.call("step", n)
            for f in ("tr_arr", "UV", "eta_n", "hnode"):
                ok, msg = compare(f, gpu.get(f, orc2.count(f)), orc2.get(f))
                assert ok, f"step {n}: {msg}"
    check_fcheck(toy_soufflet.fcheck_means(sT, sS, sU, sV, 72))
