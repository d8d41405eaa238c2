#!/usr/bin/env python3
"""Write a 1-rank partition directory `dist_1/` for a FESOM2 mesh directory.

Test infrastructure (oracle harness).  The reference's partitioner refuses
npes<2 (fvom_init.F90:1521-1525), so the trivial partition is written by hand in
the format read at oce_mesh.F90:199-256,568-663 (writer: oce_local.F90:162-317):
rank 0 owns every node/element/edge in global order, no halo, no neighbours.
"""
import sys, os


def _wrap(vals, per=6):
    out = []
    for i in range(0, len(vals), per):
        out.append(" ".join(f"{v:11d}" for v in vals[i:i + per]))
    return "\n".join(out) if out else ""


def make_dist1(meshdir):
    n2 = int(open(os.path.join(meshdir, "nod2d.out")).readline().split()[0])
    e2 = int(open(os.path.join(meshdir, "elem2d.out")).readline().split()[0])
    d2 = int(open(os.path.join(meshdir, "edgenum.out")).readline().split()[0])
    d = os.path.join(meshdir, "dist_1")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "rpart.out"), "w") as f:
        f.write(f"{1:12d}\n{n2:12d}\n")
        f.write("\n".join(f"{i:12d}" for i in range(1, n2 + 1)) + "\n")
    with open(os.path.join(d, "my_list00000.out"), "w") as f:
        f.write(f"{0:12d}\n{n2:12d}\n{0:12d}\n{_wrap(list(range(1, n2 + 1)))}\n")
        f.write(f"{e2:12d}\n{0:12d}\n{0:12d}\n{_wrap(list(range(1, e2 + 1)))}\n")
        f.write(f"{d2:12d}\n{0:12d}\n{_wrap(list(range(1, d2 + 1)))}\n")
    with open(os.path.join(d, "com_info00000.out"), "w") as f:
        f.write(f"{0:12d}\n")
        for _ in range(3):          # com_nod2D, com_elem2D, com_elem2D_full
            for _ in range(2):      # receive side, send side
                f.write(f"{0:12d}\n\n{1:12d}\n\n")   # PEnum, PE(1:0), ptr(1:1), list(empty)
    return d


if __name__ == "__main__":
    print(make_dist1(sys.argv[1]))
