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


def test_bench_control_path_two_ranks(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per 'GPU'), with --stub-mapper so that it
    runs on CPU over gloo: the tested code is the executed code -- shard_range for the read shards, rank 0 builds the genome and
    hands it to rank 1 through files, reduce_stats is the one stats collective, rank 0 prints the one JSON line."""
    import json
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--stub-mapper", "--genome-mbp", "2", "--reads-per-step", "20000"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["unit"] == "reads/s"
    st = j["stats_allreduce"]
    from nextgenmap_amd.sharding import STAT_NAMES
    assert tuple(st.keys()) == STAT_NAMES + ("ranks_seen",) and st["ranks_seen"] == 2
    # 40 000 reads over both ranks; the stub leaves global reads 999, 1999, ... unmapped
    assert st["reads"] == 40000 and st["unmapped"] == 40 and st["mapped"] == 40000 - 40 and st["written"] == 40000
    assert st["pairs_total"] == 20000 and st["insert_cnt"] == 20000 - 40
    assert abs(j["value"] - 40000 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    assert j["cpu_baseline"] is None and j["end_to_end"] is None
