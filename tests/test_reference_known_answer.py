"""CPU: pins the oracle chain to the reference's OWN known answers.  The reference's CI test test_souf
(setups/test_souf/setup.yml:82-88, `fcheck` block) is reproduced by the reference binary built here from the
reference's sources (oracle/_ref, test infrastructure): 8 ranks, 72 steps, Soufflet channel.  Skipped where the
binary or mpiexec is unavailable.  The committed record of the run made in the build container is checked always."""
import json
import os
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KA = json.load(open(os.path.join(REPO, "tests", "golden", "known_answers.json")))
TOL = {"temp": 1e-13, "sst": 1e-13, "salt": 0.0, "u": 1e-9, "v": 5e-9}      # u, v: gfortran-vs-flang round-off


def _check(means):
    for k, ref in KA["fcheck"].items():
        assert abs(means[k] - ref) <= TOL[k] * max(abs(ref), 1e-300) + (0 if TOL[k] else 0), (k, means[k], ref)


def test_committed_record_matches_fcheck():
    _check(KA["reference_built_here"])


def test_reference_binary_reproduces_fcheck():
    exe = os.path.join(REPO, "oracle", "_ref", "fesom_oracle.x")
    if not (os.path.exists(exe) and os.path.exists("/opt/conda/bin/mpiexec")):
        pytest.skip("reference binary / mpiexec not available")
    from oracle.ref import run_ref
    rd, rc, lines = run_ref.run("souf", 8, 72, mode="step", mean=True, dump_mesh=False)
    if rc != 0:
        pytest.skip(f"mpiexec could not run here (rc={rc})")
    means = {l.split()[1]: float(l.split()[2]) for l in lines if l.startswith("ORACLE_MEAN")}
    _check(means)
