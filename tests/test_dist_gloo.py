"""CPU: the N>1 contract of bench.py (one process per rank, barrier-bracketed timed region, MAX over ranks, aggregate
value) exercised with world_size=2 on the gloo backend.  The data path has no collective in this round (replicas)."""
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
