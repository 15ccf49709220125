"""N > 1 path on CPU: two gloo ranks exercise the sharding arithmetic and the episode-return
all-gather that bench.py and a training loop use on a GPU node (there the backend is RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from pcc_rl_amd import distributed as D
    rank, world, local = D.init_process_group(backend="gloo")
    assert world == 2 and rank in (0, 1)
    n = 5
    base = D.env_gid_base(rank, n)
    assert base == rank * n
    # each rank's "episode returns" are its global env ids: the gathered vector must be 0..2n-1
    mine = torch.arange(base, base + n, dtype=torch.float32)
    allr = D.gather_episode_returns(mine)
    assert allr.tolist() == [float(i) for i in range(world * n)], allr
    buf = torch.empty(world * n, dtype=torch.float32)
    assert D.gather_episode_returns(mine, out=buf) is buf and buf.tolist() == allr.tolist()
    assert D.max_over_ranks(1.0 + rank) == 2.0
    assert D.sum_over_ranks(10.0 * (rank + 1)) == 30.0
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


GATHER4_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    from pcc_rl_amd import distributed as D
    rank, world, local = D.init_process_group(backend="gloo")
    assert world == 4
    # ranks whose envs finish their episodes at different steps: a rank gathers when ITS episodes end, every rank takes part
    # in every gather (bench.py gathers at the same step count on every rank: episodes have one length) -- here the returns a
    # rank contributes at gather g are those of its envs that have finished by then, -1 for the others
    n, T = 6, 12
    finish = [(3 + (rank * n + i) %% 5) for i in range(n)]            # env i of this rank finishes at this step
    info = D.collective_info()
    assert info["backend"] == "gloo" and info["dist_world_size"] == 4
    for g, t in enumerate((4, 8, 12)):
        mine = torch.tensor([float(100 * rank + i) if finish[i] <= t else -1.0 for i in range(n)])
        allr = D.gather_episode_returns(mine)
        assert allr.numel() == world * n
        for r in range(world):
            for i in range(n):
                want = float(100 * r + i) if (3 + (r * n + i) %% 5) <= t else -1.0
                assert allr[r * n + i].item() == want, (g, r, i)
    assert D.gather_floats(1.5 * rank) == [0.0, 1.5, 3.0, 4.5]
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_four_rank_gather_with_uneven_finish_steps(tmp_path):
    script = tmp_path / "worker4.py"
    script.write_text(GATHER4_WORKER)
    port = free_port()
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="4", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_single_process_is_identity():
    import torch
    from pcc_rl_amd import distributed as D
    assert D.rank_info()[1] >= 1
    x = torch.arange(4.0)
    assert D.gather_episode_returns(x) is x
    assert D.max_over_ranks(3.5) == 3.5 and D.sum_over_ranks(2.0) == 2.0


# ---- the same path on a GPU box: two ranks of the SIMULATOR (sharing cuda:0, gloo between them) ----
import json

import pytest

SHARD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import pcc_rl_amd
    from pcc_rl_amd import distributed as D
    rank, world, local = D.init_process_group(backend="gloo")
    dev = torch.device("cuda:0")
    n, T = 320, 25
    # rank r owns the global env ids [r*n, (r+1)*n): same seed, its own id base
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=dev, seed=9, env_gid_base=D.env_gid_base(rank, n), record_steps=True,
                                       auto_reset=True, max_steps=10)
    gen = torch.Generator().manual_seed(77)
    acts = torch.rand((T, world * n), generator=gen, dtype=torch.float64) * 2 - 1   # actions by GLOBAL env id
    env.reset()
    rows = []
    for t in range(T):
        o, r, d, info = env.step(acts[t, rank * n:(rank + 1) * n].to(dev))
        rows.append(info["steps"].clone())
    mine = torch.stack(rows, 1).cpu()                              # [n, T, 19]
    ret = D.gather_episode_returns(env.episode_returns().to(torch.float32).cpu())
    gathered = [torch.empty_like(mine) for _ in range(world)]
    torch.distributed.all_gather(gathered, mine)
    if rank == 0:
        # the same 2n envs as ONE batch on one handle: shard r must equal rows [r*n, (r+1)*n)
        big = pcc_rl_amd.BatchedNetworkEnv(world * n, device=dev, seed=9, record_steps=True, auto_reset=True, max_steps=10)
        big.reset()
        rows = []
        for t in range(T):
            o, r, d, info = big.step(acts[t].to(dev))
            rows.append(info["steps"].clone())
        whole = torch.stack(rows, 1).cpu()
        assert torch.equal(torch.cat(gathered, 0), whole), "sharded results differ from the single batch"
        assert torch.equal(ret, big.episode_returns().to(torch.float32).cpu()), "gathered episode returns differ"
        print("shards ok")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""") % ROOT


@pytest.mark.gpu
def test_two_simulator_ranks_equal_one_batch(tmp_path):
    """Rank r of a 2-rank job holds exactly the envs [r*n, (r+1)*n) of a single 2n-env batch (results
    and gathered episode returns), i.e. sharding over GPUs never changes a result."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    assert "shards ok" in outs[0]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` alone (no torchrun) runs two ranks and says so in its line; with short
    episodes the per-episode all-gather of the returns shows up too."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo",
           "--envs", "1024", "--steps", "30", "--warmup", "5", "--repeats", "1", "--max-steps", "10", "--no-cpu-baseline"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stdout
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["episode_return_allgathers"] >= 3
    assert out["value"] > 0 and out["steps"] == 30
    # the line proves its own rank count: what the process group reports, and every rank's own clock
    assert out["distributed"]["dist_world_size"] == 2 and out["distributed"]["backend"] == "gloo"
    assert len(out["distributed"]["per_rank_ms_per_step"]) == 2


@pytest.mark.gpu
def test_one_rank_through_torchrun_equals_the_plain_run():
    """`--gpus 1` launched the way the driver launches N ranks (torch.distributed.run, RCCL process group of one rank)
    must measure what the plain single-process run measures: same value within a few per cent, and the line says which
    launcher and backend it went through."""
    common = ["--gpus", "1", "--steps", "400", "--warmup", "20", "--repeats", "3", "--no-cpu-baseline", "--no-pmc", "--no-policy"]
    envv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(cmd):
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=envv)
        assert res.returncode == 0, res.stdout[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])

    # (the fastest of each command's three runs is compared: a box in the middle of a ten-minute test session has runs that
    # differ by a few per cent among themselves -- the full suite once saw the medians 5 % apart -- and one retry of the pair)
    for attempt in range(2):
        plain = run([sys.executable, os.path.join(ROOT, "bench.py")] + common)
        tr = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + common)
        assert plain["distributed"]["launcher"] == "plain" and tr["distributed"]["launcher"] == "torchrun"
        assert tr["n_gpus"] == 1 and tr["distributed"]["dist_world_size"] == 1
        ratio = min(plain["runs_ms_per_step"]) / min(tr["runs_ms_per_step"])
        if abs(ratio - 1.0) < 0.05:
            break
    assert abs(ratio - 1.0) < 0.05, (tr["value"], plain["value"], tr["runs_ms_per_step"], plain["runs_ms_per_step"])


def test_bench_refuses_a_rank_count_mismatch():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 8" in res.stdout


RCCL_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # torch's "nccl" backend is RCCL on ROCm
    n = 65536
    mine = torch.arange(n, dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, mine)             # the episode-return exchange of bench.py / distributed.py
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the bench's max-over-ranks
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(out, mine) and float(t.item()) == 3.5
    dist.destroy_process_group()
    print("rccl ok")
""") % ROOT


import pytest  # noqa: E402


@pytest.mark.gpu
def test_rccl_backend_runs_the_exchange_on_one_rank():
    """A node with several GPUs is not available to the tests; this at least executes the RCCL backend on the GPU box --
    communicator set-up and the two collectives of the multi-GPU path (all-gather of episode returns, MAX all-reduce of the
    bench clock) -- with a world of one rank."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", RCCL_WORKER], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stderr[-2000:]
