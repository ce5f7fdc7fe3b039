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


def spawn_ranks(nranks, argv, env_extra=None, timeout=None):
    """Run `python argv...` as `nranks` fresh child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set as torch.distributed.run sets them), the way the reference starts one
    pipeline process per channel block (lwa352-start-pipeline.sh:1-8).  The caller must not have touched the
    GPU: the children are new processes (no fork of a HIP context, no exec of a process that holds one).
    Returns (exit code, stdout of rank 0, list of per-rank stderr tails); rank 0 prints the job's result."""
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, text=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE))
    rc, out0, errs = 0, "", []
    for r, pr in enumerate(procs):
        try:
            o, e = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = pr.communicate()
            rc = rc or 124
        if r == 0:
            out0 = o or ""
        errs.append((e or "")[-2000:])
        rc = rc or pr.returncode
    return rc, out0, errs
