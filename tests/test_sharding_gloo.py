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


FAKE_NGM_HIP = textwrap.dedent('''\
    #!%s
    """stand-in for `ngm-hip -g a,b --shard-output` (no GPU here): what bench.py's config-4 leg reads from the real program -- the
    per-shard log lines on stderr and one SAM record per input record in the output file"""
    import sys
    a = sys.argv[1:]
    out = a[a.index("-o") + 1]
    gpus = a[a.index("-g") + 1].split(",")
    assert "--shard-output" in a
    files = [a[a.index(f) + 1] for f in ("-1", "-2") if f in a] or [a[a.index("-q") + 1]]
    n = sum(sum(1 for _ in open(f)) // 4 for f in files)
    per = n // len(gpus)
    with open(out, "w") as f:
        f.write("@HD\\tVN:1.0\\n")
        for i in range(n):
            f.write("r%%d\\t4\\t*\\t0\\t0\\t*\\t*\\t0\\t0\\tA\\tI\\n" %% i)
    for k in range(len(gpus)):
        sys.stderr.write("[PREPROCESS] Reference and index ready: 0.%%d00 s (cache)\\n" %% (k + 1))
        sys.stderr.write("[MAIN] Done (%%d reads mapped (99.00%%%%), %%d reads not mapped, %%d lines written)\\n" %% (per - 1, 1, per))
        sys.stderr.write("[MAIN] Input to output: 0.%%d50 s (estimation pass + mapping pass, first input byte to output closed)\\n" %% (k + 1))
    sys.stderr.write("[MAIN] Done, %%d shards summed (%%d reads mapped (99.00%%%%), %%d reads not mapped, %%d lines written; %%d reads; %%d pairs with both mates mapped, %%d of them broken, mean insert size 350.0)\\n"
                     %% (len(gpus), len(gpus) * (per - 1), len(gpus), len(gpus) * per, len(gpus) * per, len(gpus) * per // 2 - 1, 3))
    sys.stderr.write("[MAIN] %%d shards appended to the output in 0.010 s\\n" %% len(gpus))
''')


def test_bench_config4_leg_two_ranks(tmp_path):
    """bench.py --gpus 2: after the weak-scaling loop rank 0 runs the product once over ONE input, sharded over both 'GPUs'
    (`ngm-hip -g 0,1 --shard-output`, here a stand-in program: no GPU), and reports it as `end_to_end` with "scaling": "strong"."""
    import json
    fake = tmp_path / "fake-ngm-hip"
    fake.write_text(FAKE_NGM_HIP % sys.executable)
    fake.chmod(0o755)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--stub-mapper", "--genome-mbp", "2", "--reads-per-step", "4096", "--e2e-reads", "10000", "--ngm-hip-exe", str(fake)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "weak" and j["n_gpus"] == 2
    e = j["end_to_end"]
    assert "error" not in e, e
    assert e["scaling"] == "strong" and e["shards"] == 2 and e["reads"] == 10000 and e["sam_records"] == 10000
    assert e["per_shard_input_to_output_s"] == [0.15, 0.25] and e["per_shard_index_load_s"] == [0.1, 0.2] and e["append_s"] == 0.01
    assert abs(e["seconds_first_input_byte_to_concatenated_sam_closed"] - 0.26) < 1e-9 and abs(e["value"] - 10000 / 0.26) < 1e-6
    assert e["stats_summed_over_shards"] == {"mapped": 2 * 4999, "unmapped": 2, "written": 10000}
    assert e["stats_summed_by_the_parent_process"] == {"shards": 2, "mapped": 2 * 4999, "unmapped": 2, "written": 10000, "reads": 10000, "pairs_total": 4999, "pairs_broken": 3, "mean_insert_size": 350.0}
    assert "-g 0,1 --shard-output" in e["command"]
