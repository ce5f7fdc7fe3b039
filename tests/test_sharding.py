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


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus N` on its own (no torch.distributed.run, no WORLD_SIZE) starts N ranks itself and prints
    one line with n_gpus = N whose time is the max over ranks; under torch.distributed.run the same flag is only checked
    against WORLD_SIZE.  --selftest-spawn replaces the GPU work by a fixed per-rank time (rank r: (r+1) ms)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                        "--selftest-spawn"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["nchan_total"] == 192
    assert abs(res["ms_per_step"] - 0.2) < 1e-9          # max over ranks = rank 1's 2 ms over 10 steps
    # every rank's own time travels in the line, and every rank was pinned to its own share of the host's cores
    assert res["per_rank_ms"] == [0.1, 0.2] and res["per_rank_ms_min"] == 0.1 and res["per_rank_ms_max"] == 0.2
    allowed = sorted(os.sched_getaffinity(0))
    m0, m1 = res["rank_cpus"]
    if len(allowed) >= 2:
        assert m0 and m1 and not (set(m0) & set(m1)) and sorted(m0 + m1) == allowed and res["placement"] == "share"
        assert max(m0) < min(m1)                         # contiguous slices in rank order
    # the driver's launch line gives the same answer
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
           "--selftest-spawn"]
    r2 = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    res2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][0])
    assert res2["n_gpus"] == 2 and res2["ms_per_step"] == res["ms_per_step"]
    # a mismatch between --gpus and the launcher's world size is an error, not a silent 1-rank run
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-spawn"], cwd=ROOT,
                        env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r3.returncode != 0


@pytest.mark.parametrize("workload", ["config5", "config5_blocks"])
def test_bench_workload_config5_on_the_spawn_path(workload):
    """`bench.py --gpus 2 --workload config5` (round 4: BASELINE config 5 as N ranks run it; lwa352-start-pipeline.sh:1-8 starts
    one pipeline process per channel block): the flag travels to every rank the launcher starts, each rank takes its own 96
    channels, and rank 0's line names the workload -- on the --selftest-spawn path (gloo, two ranks, no GPU work)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "10",
                        "--warmup", "1", "--selftest-spawn"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["config"]["workload"] == "selftest:" + workload
    assert res["config"]["nchan_total"] == 192 and res["config"]["chan0_per_rank"] == [0, 96]
    assert res["per_rank_ms"] == [0.1, 0.2]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "nonsense", "--selftest-spawn"], cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0


def test_pin_rank_follows_sysfs_and_falls_back(tmp_path):
    """pin_rank takes the cores sysfs names for the GPU's PCI device (numa_node / local_cpulist) and, where the platform
    says nothing (numa_node -1, no such device), the rank's own share of the allowed CPUs.  Runs in a child process: the
    affinity mask of the test runner is left alone."""
    allowed = sorted(os.sched_getaffinity(0))
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    near = allowed[-2:] if len(allowed) >= 2 else allowed
    (dev / "numa_node").write_text("1\n")
    (dev / "local_cpulist").write_text(",".join(str(c) for c in near) + "\n")
    bad = tmp_path / "0000:05:00.0"
    bad.mkdir()
    (bad / "numa_node").write_text("-1\n")
    (bad / "local_cpulist").write_text("\n")
    code = ("import json, sys; sys.path.insert(0, %r); import caltech_bifrost_dsp_amd\n"
            "from caltech_bifrost_dsp_amd import sharding\n"
            "print(json.dumps(sharding.pin_rank(int(sys.argv[1]), 4, sys.argv[2] or None, %r)))" % (ROOT, str(tmp_path)))

    def run(local_rank, bus):
        r = subprocess.run([sys.executable, "-c", code, str(local_rank), bus], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    a = run(0, "0000:c1:00.0")
    assert a["source"] == "sysfs" and a["numa_node"] == 1 and a["cpus"] == near
    for bus in ("0000:05:00.0", "0000:ff:00.0", ""):
        b = run(1, bus)
        assert b["source"] == "share" and b["numa_node"] is None
        assert set(b["cpus"]) == sharding.rank_cpu_share(1, 4, set(allowed))
    shares = [sharding.rank_cpu_share(r, 4, set(range(16))) for r in range(4)]
    assert shares == [set(range(0, 4)), set(range(4, 8)), set(range(8, 12)), set(range(12, 16))]
    assert sharding.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
