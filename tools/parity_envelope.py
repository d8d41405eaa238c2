#!/usr/bin/env python3
"""Free-running agreement with the reference over the reference CI's run length (setups/test_pi/setup.yml:12: one day = 96
steps), put next to the reference's OWN reproducibility: the reference is not bit-reproducible across partitions and its pARMS
solve stops at ||scaled residual|| < 1e-10 from a partition-dependent iterate.  Three runs of pi with the default
physics (KPP + GM + Redi, analytic surface forcing), same namelists / initial state / forcing:
    R2 = reference, 2 MPI ranks      R8 = reference, 8 MPI ranks      G = the reference's set-up stepping on the GPU (drop-in, 1 rank)
and max |difference| of eta_n, T, S, U, hnode after NSTEPS (default 20 and 96): |G - R2|, |G - R8| against |R8 - R2|.
usage: parity_envelope.py [out.json] [nsteps,nsteps,...]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import numpy as np
from oracle.ref import run_ref
from oracle.ref.compare_oracle import assemble
from refdump import read_dump


def state(cfg, ranks, n, mode, exe):
    rd, rc, lines = run_ref.run(cfg, ranks, n, mode=mode, dump=(n,), exe_name=exe)
    assert rc == 0, open(os.path.join(rd, "stdout.log")).read()[-2000:]
    s = [read_dump(os.path.join(rd, "dumps", f"setup.r{r:05d}.bin")) for r in range(ranks)]
    d = [read_dump(os.path.join(rd, "dumps", f"state{n:04d}.r{r:05d}.bin")) for r in range(ranks)]
    out = {f: assemble(d, s, f) for f in ("eta_n", "tr_arr", "UV", "hnode")}
    vol = out["hnode"] * assemble(s, s, "areasvol")[:, :-1]
    out["content"] = [float((out["tr_arr"][k] * vol).sum()) for k in (0, 1)]
    its = [l for l in lines if "ITER" in l.upper()]
    return out, its


def main():
    outfn = sys.argv[1] if len(sys.argv) > 1 else None
    steps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [20, 96]
    cfg = os.environ.get("ENVELOPE_CFG", "pi_default")
    os.environ["FESOM_GPU_DEVICE"] = "0"
    res = {"cfg": cfg, "note": __doc__.split("usage")[0].strip(), "steps": {}}
    for n in steps:
        G, _ = state(cfg, 1, n, "gpu", "fesom_gpu_dropin.x")
        R2, _ = state(cfg, 2, n, "step", "fesom_oracle.x")
        R8, _ = state(cfg, 8, n, "step", "fesom_oracle.x")
        row = {}
        for name, a, b in (("G-R2", G, R2), ("G-R8", G, R8), ("R8-R2", R8, R2)):
            row[name] = {"eta_n": float(np.abs(a["eta_n"] - b["eta_n"]).max()), "T": float(np.abs(a["tr_arr"][0] - b["tr_arr"][0]).max()),
                         "S": float(np.abs(a["tr_arr"][1] - b["tr_arr"][1]).max()), "UV": float(np.abs(a["UV"] - b["UV"]).max()),
                         "hnode": float(np.abs(a["hnode"] - b["hnode"]).max()),
                         "heat_content_rel": abs(a["content"][0] - b["content"][0]) / abs(b["content"][0]),
                         "salt_content_rel": abs(a["content"][1] - b["content"][1]) / abs(b["content"][1])}
        res["steps"][str(n)] = row
        print(n, json.dumps(row), flush=True)
    if outfn:
        json.dump(res, open(outfn, "w"), indent=1)


if __name__ == "__main__":
    main()
