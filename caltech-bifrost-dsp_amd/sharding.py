"""Frequency-channel sharding across the GPUs of a node.

The reference's only parallelism is channel sharding across independent pipeline processes
(32 pipelines x 96 channels: lwa352-pipeline.py:165-166, lwa352-start-pipeline.sh:1-8); every
channel is an independent correlation / beamformer batch entry, so there is no data-path
collective.  One process per GPU; rank r owns channels
[r*nchan_per_gpu, (r+1)*nchan_per_gpu).  The process group, when there is one, is used only
for barriers and for reducing timings -- never for data.
"""
import os
import socket
import subprocess
import sys

CHAN_BW_HZ = 196e6 / 8192      # 23925.78125 (capture_block.py:165)


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process: 0,0,1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_channels(nchan_total, world_size, rank):
    """-> (chan0, nchan) of `rank`.  Channels must divide evenly (768 = 8 x 96 in the design)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d of %d" % (rank, world_size))
    if nchan_total % world_size:
        raise ValueError("%d channels do not shard evenly over %d GPUs" % (nchan_total, world_size))
    n = nchan_total // world_size
    return rank * n, n


def shard_header(base_hdr, chan0, nchan, chan_bw_hz=CHAN_BW_HZ):
    """Sequence header of one shard: chan0 / sfreq / bw_hz as capture_block.py:267-272 derives them."""
    hdr = dict(base_hdr)
    hdr.update({'chan0': chan0, 'nchan': nchan, 'sfreq': chan0 * chan_bw_hz, 'bw_hz': nchan * chan_bw_hz})
    return hdr


def shard_seed(base_seed, rank):
    """Synthetic-input seed of a shard (seed 0xdeadbeef + g)."""
    return (base_seed + rank) & 0xFFFFFFFF


def init_process_group(backend="gloo"):
    """Join the job's process group if there is more than one rank; returns torch.distributed or None.
    PyTorch is plumbing here (barrier / reduce of scalars); the data path never touches it."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def max_over_ranks(dist, value):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11} (the format of sysfs cpulist files)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(pci_bus_id, sysfs_root="/sys/bus/pci/devices"):
    """(numa_node, cpus next to the device) from sysfs, or (None, None) when the platform does not say
    (no such device directory, numa_node -1, empty list: containers and single-node hosts)."""
    d = os.path.join(sysfs_root, pci_bus_id or "")
    try:
        with open(os.path.join(d, "numa_node")) as fh:
            node = int(fh.read().strip())
        with open(os.path.join(d, "local_cpulist")) as fh:
            cpus = parse_cpulist(fh.read())
    except (OSError, ValueError):
        return None, None
    if node < 0 or not cpus:
        return None, None
    return node, cpus


def rank_cpu_share(local_rank, local_world, allowed):
    """Fallback placement: the `local_rank`-th of `local_world` contiguous, disjoint slices of the allowed CPUs."""
    cpus = sorted(allowed)
    n = len(cpus)
    if local_world <= 1 or n < local_world:
        return set(cpus)
    lo, hi = (local_rank * n) // local_world, ((local_rank + 1) * n) // local_world
    return set(cpus[lo:hi])


def pin_rank(local_rank, local_world, pci_bus_id=None, sysfs_root="/sys/bus/pci/devices"):
    """Pin this process (and the block threads it starts) to the host cores next to its GPU -- the reference pins every
    block thread to a core of the GPU's socket (corr_block.py:336-338, lwa352-pipeline.py `--cores`).  The cores come from
    the device's sysfs entry (numa_node / local_cpulist); where the platform has none, every rank takes its own contiguous
    share of the allowed CPUs so that ranks do not migrate over each other.  Returns what was done."""
    if not hasattr(os, "sched_setaffinity"):
        return {"source": "unsupported", "cpus": [], "numa_node": None}
    allowed = os.sched_getaffinity(0)
    node, cpus = gpu_local_cpus(pci_bus_id, sysfs_root) if pci_bus_id else (None, None)
    source = "sysfs"
    if cpus:
        cpus &= allowed
    if not cpus:
        node, cpus, source = None, rank_cpu_share(local_rank, local_world, allowed), "share"
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return {"source": "failed", "cpus": sorted(allowed), "numa_node": node}
    return {"source": source, "cpus": sorted(os.sched_getaffinity(0)), "numa_node": node, "pci_bus_id": pci_bus_id}


def gather_over_ranks(dist, value):
    """Every rank's scalar, in rank order, on every rank ([value] without a process group)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def spawn_ranks(nranks, argv, env_extra=None, timeout=1800):
    """Run `python argv...` as `nranks` fresh child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set as torch.distributed.run sets them), the way the reference starts one
    pipeline process per channel block (lwa352-start-pipeline.sh:1-8).  The caller must not have touched the
    GPU: the children are new processes (no fork of a HIP context, no exec of a process that holds one).
    Every rank pins itself to its GPU's cores (pin_rank).  stderr of every rank goes to a temporary file (a pipe
    that nobody drains would block a chatty rank, and rank 0 with it at the next barrier).
    Returns (exit code, stdout of rank 0, list of per-rank stderr tails); rank 0 prints the job's result."""
    import tempfile
    import time
    port = free_port()
    procs, errfiles = [], []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(env_extra or {})
        ef = tempfile.TemporaryFile(mode="w+")
        errfiles.append(ef)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, text=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    rc, out0 = 0, ""
    deadline = None if timeout is None else time.time() + timeout
    try:
        out0 = procs[0].communicate(timeout=timeout)[0] or ""       # (the only pipe: drained while waiting)
        for pr in procs[1:]:
            pr.wait(timeout=None if deadline is None else max(1.0, deadline - time.time()))
    except subprocess.TimeoutExpired:
        rc = 124
        for q in procs:
            q.kill()
        out0 = out0 or (procs[0].communicate()[0] or "")
        for q in procs:
            q.wait()
    errs = []
    for pr, ef in zip(procs, errfiles):
        ef.seek(0)
        errs.append(ef.read()[-2000:])
        ef.close()
        rc = rc or pr.returncode
    return rc, out0, errs
