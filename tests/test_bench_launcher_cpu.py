"""bench.py --gpus N without a launcher around it: the command starts its own N ranks, prints ONE line from
rank 0, and fails loudly — exit code != 0, nothing on stdout — when it cannot have N devices.  The device is
tests/stub_device.py here (BENCH_STUB_DEVICE=1: no GPU in this container); the same self-launch with real contexts is the
8-rank gloo dry run committed under profiles/ (r06_bench_8rank_gloo_line.json)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    env["BENCH_EXTRAS_FILE"] = os.path.join(env.get("TMPDIR", "/tmp"), f"bench_extras_test_{os.getpid()}.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, cwd=ROOT)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    K, B = 8, 6
    rc, out, err = _run(["--gpus", "2", "--nseq", str(B), "--pool", "4", "--steps", str(K), "--warmup", "4"],
                        {"BENCH_STUB_DEVICE": "1", "BENCH_BACKEND": "gloo"})
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == K and line["warmup"] == 4
    assert line["data"] == "stub" and line["invalid_as_measurement"] is True
    assert "itself" in line["launched_by"]
    c = line["config"]
    assert c["nav_gather"] == "ok" and c["nav_gather_ranks"] == 2 and c["nav_gather_records"] == 2 * K * B
    assert c["frames_per_step"] == 2 * B
    assert len(lines[0]) <= 4096


def test_gpus_2_over_rccl_on_a_box_without_two_devices_fails_loudly():
    """The default backend (RCCL, one rank per GPU) with fewer devices than ranks: no line, exit code != 0 — never n_gpus: 1."""
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {})
    assert rc != 0
    assert out.strip() == ""
    assert "--gpus 2" in err


def test_world_size_that_contradicts_gpus_fails_loudly():
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and out.strip() == "" and "WORLD_SIZE=1" in err


def test_a_rank_that_dies_stops_the_others_and_the_command_fails():
    rc, out, err = _run(["--gpus", "2", "--nseq", "4", "--pool", "4", "--steps", "6", "--warmup", "4"],
                        {"BENCH_STUB_DEVICE": "1", "BENCH_BACKEND": "gloo", "BENCH_STUB_FAIL_RANK": "1"}, timeout=240)
    assert rc != 0
    assert out.strip() == ""
    assert "rank 1 ended with exit code" in err


def test_device_side_records_equal_the_host_side_ones():
    """shard.nav_records_device (what the RCCL gather sends: column picks on the raw edgehip_nav bytes, on the device) against
    shard.nav_records on the same log — here on CPU tensors, the same code."""
    import torch
    from rebvo_amd import edgehip, shard
    K, B = 5, 7
    rs = np.random.RandomState(5)
    arr = np.zeros((K, B), dtype=edgehip.NAV_DTYPE)
    arr["frame"] = np.arange(K)[:, None]
    for f in ("kn", "klm_num"):
        arr[f] = rs.randint(0, 16000, (K, B))
    arr["estimation_ok"] = rs.randint(0, 2, (K, B))
    for f in ("Pos", "V", "W", "Pose"):
        arr[f] = rs.normal(size=arr[f].shape)
    raw = torch.from_numpy(arr.view(np.uint8).reshape(K, B, edgehip.NAV_DTYPE.itemsize).copy())
    seq_ids = list(range(40, 40 + B))
    got = shard.nav_records_device(raw, 3, seq_ids).numpy()
    assert np.array_equal(got, shard.nav_records(arr, 3, seq_ids))
