"""Multi-GPU plumbing: one process per GPU, envs split by contiguous global id ranges, no
collective inside step/reset.  The only exchange on this path is gathering the per-env episode
returns when episodes end -- RCCL over xGMI on a GPU node (torch's
"nccl" backend), gloo in the CPU tests."""
import os

import torch


def rank_info():
    """(rank, world_size, local_rank) from the torchrun environment (1 process = defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend=None, device=None):
    """Initialise torch.distributed when launched by torchrun; returns (rank, world, local_rank)."""
    rank, world, local_rank = rank_info()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend, **kwargs)
    return rank, world, local_rank


def env_gid_base(rank, envs_per_rank):
    """Global id of a rank's env 0: rank r owns ids [r*n, (r+1)*n).  The Philox streams are keyed
    by global id, so a batch gives the same results however it is sharded."""
    return int(rank) * int(envs_per_rank)


def gather_episode_returns(returns, out=None):
    """All-gather a rank's [n] episode returns into [world*n] (rank order).  Identity when not
    distributed.  One call per episode: 4 bytes x n per rank -- latency-bound by construction."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return returns if out is None else out.copy_(returns)
    world = dist.get_world_size()
    returns = returns.contiguous()
    if out is None:
        out = torch.empty((world * returns.numel(),), dtype=returns.dtype, device=returns.device)
    dist.all_gather_into_tensor(out, returns)
    return out


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float across ranks (bench timing contract)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value, device="cpu"):
    """Every rank's python float, in rank order (a list of world_size floats; [value] when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = torch.empty((dist.get_world_size(),), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu().tolist()]


def collective_info():
    """What proves which library carried the collectives and how many ranks it saw: backend name, the world size the
    process group reports, and (backend "nccl" = RCCL on ROCm) the RCCL version torch was built against."""
    import torch.distributed as dist
    info = {"backend": None, "dist_world_size": 1, "rccl_version": None}
    if dist.is_available() and dist.is_initialized():
        info["backend"] = dist.get_backend()
        info["dist_world_size"] = dist.get_world_size()
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    return info


def pin_to_gpu_numa_node(local_rank):
    """Bind this process to the host cores of the NUMA node its GPU hangs off (a rank's launches and the all-gather's
    host side stay off the inter-socket link).  Best effort: returns a description, or why nothing was done."""
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            cpulist = f.read().strip()
        cpus = set()
        for part in cpulist.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return "no local cpus for %s" % bdf
        os.sched_setaffinity(0, cpus)
        return "%s: cpus %s" % (bdf, cpulist)
    except Exception as e:   # no sysfs entry, no permission, no such attribute: run unpinned
        return "not pinned (%s)" % (type(e).__name__,)
