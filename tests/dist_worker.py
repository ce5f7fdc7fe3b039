"""Worker for tests/test_sharding.py: run under torch.distributed.run with world_size 2 (gloo).
Each rank correlates its own channel shard (through the Corr block on a CPU ring, oracle backend)
and the ranks exchange only checksums -- the data path has no collective."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import caltech_bifrost_dsp_amd  # noqa: E402,F401
from caltech_bifrost_dsp_amd import sharding  # noqa: E402
from caltech_bifrost_dsp_amd.blocks import Corr  # noqa: E402
from caltech_bifrost_dsp_amd.ring import Ring  # noqa: E402
from tests.fake_backend import OracleBackend  # noqa: E402
from tests.pipeline_util import LOG, Sink, Source, run_blocks, source_header  # noqa: E402


def main():
    out_path = sys.argv[1]
    nchan_total, nstand, ntime, gulp = 8, 8, 8, 4
    rank, local_rank, world = sharding.env_rank()
    dist = sharding.init_process_group("gloo")
    chan0, nchan = sharding.shard_channels(nchan_total, world, rank)
    # every rank can regenerate the whole band; it only processes its own channels
    full = np.random.RandomState(1234).randint(0, 255, size=(ntime, nchan_total, nstand, 2), dtype=np.uint8)
    mine = np.ascontiguousarray(full[:, chan0:chan0 + nchan])
    hdr = sharding.shard_header(source_header(nchan, nstand, 2), chan0, nchan)
    iring, oring = Ring("gpu-input"), Ring("corr-output")
    blk = Corr(LOG, iring, oring, ntime_gulp=gulp, nchan=nchan, npol=2, nstand=nstand, acc_len=ntime,
               autostartat=0, backend=OracleBackend())
    sink = Sink(oring, blk.ogulp_size)
    if dist is not None:
        dist.barrier()
    run_blocks([blk], Source(iring, [(hdr, mine, gulp * nchan * nstand * 2)]), [sink])
    (ohdr, _, spans), = sink.sequences
    digest = hashlib.sha256(spans[0].tobytes()).hexdigest()
    t = sharding.max_over_ranks(dist, float(rank + 1))
    gathered = [None] * world
    if dist is not None:
        dist.all_gather_object(gathered, {"rank": rank, "chan0": ohdr["chan0"], "nchan": ohdr["nchan"],
                                          "sfreq": ohdr["sfreq"], "sha": digest})
        dist.barrier()
    else:
        gathered = [{"rank": 0, "chan0": ohdr["chan0"], "nchan": ohdr["nchan"], "sfreq": ohdr["sfreq"], "sha": digest}]
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"world": world, "max": t, "shards": gathered}, fh)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
