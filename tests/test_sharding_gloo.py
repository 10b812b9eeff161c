"""N > 1 control path of bench.py on CPU: world size 2 over gloo.  Checks read sharding (every read
owned by exactly one rank, contiguous, PE-safe even split) and the single stats all-reduce."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from nextgenmap_amd.sharding import shard_range, reduce_stats, STAT_NAMES
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_reads = 100003
    lo, hi = shard_range(n_reads, rank, world, paired=True)
    assert lo %% 2 == 0 and (hi %% 2 == 0 or hi == n_reads)
    # every rank reports its own counts; the reduce must give the global truth on every rank
    local = {k: 0 for k in STAT_NAMES}
    local["reads"] = hi - lo
    local["mapped"] = (hi - lo) // 3
    local["insert_sum"] = sum(range(lo, hi)) %% 1000003
    tot = reduce_stats(local)
    bounds = [None] * world
    dist.all_gather_object(bounds, (lo, hi))
    assert bounds[0][0] == 0 and bounds[-1][1] == n_reads
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    assert tot["reads"] == n_reads, tot
    assert tot["mapped"] == sum((b - a) // 3 for a, b in bounds)
    # one file per rank: the ranks' stdout lines can interleave character by character
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rank%%d.ok" %% rank), "w").write("ok %%d" %% tot["reads"])
    dist.destroy_process_group()
''')


def test_two_rank_sharding_and_stats_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert [(tmp_path / ("rank%d.ok" % k)).read_text() for k in range(2)] == ["ok 100003"] * 2
