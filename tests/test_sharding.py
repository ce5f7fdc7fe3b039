"""Multi-GPU path on CPU: world_size 2 over gloo (127.0.0.1).  Channels shard with no data-path
collective; the union of the shards' outputs equals a single-process run over all channels."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import caltech_bifrost_dsp_amd  # noqa: F401
from caltech_bifrost_dsp_amd import sharding
from oracle import xeng_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels():
    assert [sharding.shard_channels(768, 8, r) for r in (0, 1, 7)] == [(0, 96), (96, 96), (672, 96)]
    assert sharding.shard_channels(96, 1, 0) == (0, 96)
    with pytest.raises(ValueError):
        sharding.shard_channels(100, 8, 0)
    with pytest.raises(ValueError):
        sharding.shard_channels(96, 2, 2)
    h = sharding.shard_header({'nchan': 768, 'seq0': 5}, 96, 96)
    assert h['chan0'] == 96 and h['nchan'] == 96 and h['seq0'] == 5
    assert abs(h['sfreq'] - 96 * 23925.78125) < 1e-6 and abs(h['bw_hz'] - 96 * 23925.78125) < 1e-6
    assert sharding.shard_seed(0xdeadbeef, 3) == 0xdeadbeef + 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path):
    out = tmp_path / "shards.json"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2 and res["max"] == 2.0
    shards = sorted(res["shards"], key=lambda s: s["rank"])
    assert [(s["chan0"], s["nchan"]) for s in shards] == [(0, 4), (4, 4)]
    assert abs(shards[1]["sfreq"] - 4 * 23925.78125) < 1e-6
    # single-process reference over all 8 channels, cut into the same shards
    full = np.random.RandomState(1234).randint(0, 255, size=(8, 8, 8, 2), dtype=np.uint8)
    for s in shards:
        part = np.ascontiguousarray(full[:, s["chan0"]:s["chan0"] + s["nchan"]])
        exp = orc.xgpu_correlate(part, 8, s["nchan"])
        assert hashlib.sha256(exp.tobytes()).hexdigest() == s["sha"]
    # channel independence: the shard results are slices of the all-channel planar buffer
    allc = orc.xgpu_correlate(full, 8, 8).reshape(2, 8, -1)
    for s in shards:
        part = orc.xgpu_correlate(np.ascontiguousarray(full[:, s["chan0"]:s["chan0"] + s["nchan"]]), 8, s["nchan"])
        assert np.array_equal(part.reshape(2, s["nchan"], -1), allc[:, s["chan0"]:s["chan0"] + s["nchan"]])
