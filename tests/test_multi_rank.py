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


def test_single_process_is_identity():
    import torch
    from pcc_rl_amd import distributed as D
    assert D.rank_info()[1] >= 1
    x = torch.arange(4.0)
    assert D.gather_episode_returns(x) is x
    assert D.max_over_ranks(3.5) == 3.5 and D.sum_over_ranks(2.0) == 2.0
