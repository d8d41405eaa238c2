"""CPU: the N>1 contract of bench.py (one process per rank, barrier-bracketed timed region, MAX over ranks, aggregate
value) exercised with world_size=2 on the gloo backend, and the halo-exchange plan of the partitioned step (send / receive lists of the two ranks
match item for item).  The partitioned data path itself (halo messages + the solver's global sums) needs the GPU: tests/test_gpu_partitioned.py."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_timing_contract(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time, json
        sys.path.insert(0, {REPO!r})
        from fesom2_amd import dist_util as du
        rank, world = du.init("gloo")
        assert world == 2
        el = du.timed_region(lambda: time.sleep(0.2 if rank == 0 else 0.5), world)
        v = du.aggregate_sypd(el / 10, world, 365 * 96)
        sys.stdout.write(json.dumps({{"rank": rank, "el": el, "v": v}}) + chr(10)); sys.stdout.flush()
        import torch.distributed as dist
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    import re
    rows = [json.loads(x) for x in re.findall(r"\{[^{}]*\}", r.stdout)]        # two ranks share one pipe: lines may interleave
    assert len(rows) == 2
    assert abs(rows[0]["el"] - rows[1]["el"]) < 1e-9 and rows[0]["el"] >= 0.5       # MAX over ranks, identical on both
    assert abs(rows[0]["v"] - 2 * 86400.0 / (365 * 96 * rows[0]["el"] / 10)) < 1e-6


def test_two_rank_halo_exchange_plan(tmp_path):
    """world_size 2, gloo, CPU only: the halo exchange plan of the partitioned path (partition + com lists from the host mesh
    layer, message layout of fesom_gpu_halo_pack: per neighbour a block, inside it field after field, [items][values]) moves
    the owners' values into every halo slot -- nodes, the small element halo and the full element halo."""
    script = tmp_path / "h.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, {REPO!r})
        from fesom2_amd.mesh import Mesh
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        m = Mesh.load(os.path.join({REPO!r}, "tests", "golden", "meshes", "pi"), dt=900.0, npes=world, mype=rank)
        part = m.part_p.contents
        g = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy() if n > 0 else np.zeros(0, dtype=np.int32)
        bad = 0
        for kind, c, glob, nloc in (("nod", part.com_nod2D, m.myList_nod2D, m.myDim_nod2D + m.eDim_nod2D),
                                    ("elem", part.com_elem2D, m.myList_elem2D, m.myDim_elem2D + m.eDim_elem2D),
                                    ("elem_full", part.com_elem2D_full, m.myList_elem2D, len(m.myList_elem2D))):
            rPE, rptr, sPE, sptr = g(c.rPE, c.rPEnum), g(c.rptr, c.rPEnum + 1), g(c.sPE, c.sPEnum), g(c.sptr, c.sPEnum + 1)
            rlist, slist = g(c.rlist, rptr[-1] - 1) - 1, g(c.slist, sptr[-1] - 1) - 1
            W = (3, 2)                                         # two fields: 3 and 2 values per item
            val = lambda gid, k, w: gid * 10.0 + k + 0.1 * w    # what the owner holds
            fields = [np.full((nloc, w), -1.0) for w in W]
            own = len(glob) if False else (m.myDim_nod2D if kind == "nod" else m.myDim_elem2D)
            for k, f in enumerate(fields):
                for w in range(W[k]):
                    f[:own, w] = val(glob[:own].astype(np.float64), k, w)
            Wt = sum(W)
            send, recv = np.zeros((sptr[-1] - 1) * Wt), np.zeros((rptr[-1] - 1) * Wt)
            for p in range(len(sPE)):
                first, cnt = sptr[p] - 1, sptr[p + 1] - sptr[p]; off = 0
                for k, f in enumerate(fields):
                    send[first * Wt + cnt * off: first * Wt + cnt * (off + W[k])] = f[slist[first:first + cnt]].ravel(); off += W[k]
            ops = []
            ts, tr = torch.from_numpy(send), torch.from_numpy(recv)
            for p in range(len(rPE)):
                ops.append(dist.P2POp(dist.irecv, tr[(rptr[p] - 1) * Wt:(rptr[p + 1] - 1) * Wt], int(rPE[p])))
            for p in range(len(sPE)):
                ops.append(dist.P2POp(dist.isend, ts[(sptr[p] - 1) * Wt:(sptr[p + 1] - 1) * Wt], int(sPE[p])))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for p in range(len(rPE)):
                first, cnt = rptr[p] - 1, rptr[p + 1] - rptr[p]; off = 0
                for k, f in enumerate(fields):
                    f[rlist[first:first + cnt]] = recv[first * Wt + cnt * off: first * Wt + cnt * (off + W[k])].reshape(cnt, W[k]); off += W[k]
            for k, f in enumerate(fields):
                for w in range(W[k]):
                    exp = val(glob[:nloc].astype(np.float64), k, w)
                    sel = np.zeros(nloc, bool); sel[:own] = True; sel[rlist] = True      # owned + every slot of this halo kind
                    bad += int((f[sel, w] != exp[sel]).sum())
        sys.stdout.write("HALO " + json.dumps({{"rank": rank, "bad": bad}}) + chr(10)); sys.stdout.flush()
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    import json, re
    rows = [json.loads(x) for x in re.findall(r"HALO (\{[^{}]*\})", r.stdout)]
    assert len(rows) == 2 and all(x["bad"] == 0 for x in rows), rows
