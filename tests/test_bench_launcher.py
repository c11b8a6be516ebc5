"""bench.py as its own rank launcher (the role of the reference's per-device process pool, inference.py:466-488), on the CPU:
`--gpus N` without WORLD_SIZE starts N fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, their sample slices
tile the job (weak: N x samples, strong: samples split as [r*S/N, (r+1)*S/N)), a failing rank makes the parent fail, and a
WORLD_SIZE that contradicts --gpus is refused.  (`--dry-run-ranks`: the ranks report and exit before anything touches a GPU.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, "--dry-run-ranks"] + args, env=env, capture_output=True, text=True, timeout=120)


def _lines(out):
    return sorted((json.loads(l) for l in out.splitlines() if l.startswith("{")), key=lambda d: d["rank"])


def test_parent_starts_one_process_per_gpu_weak_and_strong():
    r = _run(["--gpus", "4", "--scaling", "weak"])
    assert r.returncode == 0, r.stderr
    ranks = _lines(r.stdout)
    assert [d["rank"] for d in ranks] == [0, 1, 2, 3] and all(d["world"] == 4 and d["local_rank"] == d["rank"] for d in ranks)
    assert len({d["port"] for d in ranks}) == 1 and all(d["master"] == "127.0.0.1" for d in ranks)
    assert [d["slice"] for d in ranks] == [[0, 40], [40, 80], [80, 120], [120, 160]] and ranks[0]["samples_total"] == 160
    # the default for several ranks: the STRONG split of one complex's 40 samples (BASELINE configs[3]) is the headline
    r = _run(["--gpus", "8"])
    assert r.returncode == 0, r.stderr
    ranks = _lines(r.stdout)
    assert [d["slice"] for d in ranks] == [[5 * i, 5 * i + 5] for i in range(8)] and ranks[0]["samples_total"] == 40
    assert all(d["scaling"] == "strong" for d in ranks)
    r = _run(["--gpus", "1"])
    assert _lines(r.stdout)[0]["slice"] == [0, 40] and _lines(r.stdout)[0]["scaling"] == "weak"      # the N = 1 line is unchanged
    r = _run(["--gpus", "3", "--scaling", "strong"])                  # uneven split: [0,13) [13,26) [26,40)
    assert [d["slice"] for d in _lines(r.stdout)] == [[0, 13], [13, 26], [26, 40]]


def test_a_failing_rank_fails_the_parent():
    r = _run(["--gpus", "2"], {"DDP_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0 and "rank 1 exited with code 3" in r.stderr


def test_rank_of_an_external_launcher_and_world_size_mismatch():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, drop=())      # as under torch.distributed.run
    got = _lines(r.stdout)
    assert r.returncode == 0 and len(got) == 1
    plan = got[0].pop("hbm_plan")
    assert got == [{"rank": 1, "local_rank": 1, "world": 2, "master": os.environ.get("MASTER_ADDR"), "port": os.environ.get("MASTER_PORT"),
                    "samples_total": 40, "slice": [20, 40], "scaling": "strong", "device": "cuda:1"}] and plan["samples"] == 20
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_eight_ranks_bind_their_own_gpu_and_size_memory_per_shard():
    """8-GPU readiness without the node: under `--gpus 8` every child would bind cuda:LOCAL_RANK, the
    strong split gives each 5 of the 40 samples, and the big per-layer arrays (G, messages) are sized for the SHARD - 1/8 of the
    40-sample job's 5 GB - far inside one GPU's 288 GB; the weak run (40 samples per rank) needs the single-GPU job's memory."""
    r = _run(["--gpus", "8"])
    assert r.returncode == 0, r.stderr
    ranks = _lines(r.stdout)
    assert [d["device"] for d in ranks] == [f"cuda:{i}" for i in range(8)]
    one = _lines(_run(["--gpus", "1"]).stdout)[0]["hbm_plan"]
    for d in ranks:
        p = d["hbm_plan"]
        assert p["samples"] == 5 and p["g_and_messages_bytes_largest_layer"] * 8 == one["g_and_messages_bytes_largest_layer"]
        assert p["g_and_messages_bytes_largest_layer"] < 0.01 * p["hbm_bytes_per_gpu"]
    assert 4e9 < one["g_and_messages_bytes_largest_layer"] < 8e9
    weak = _lines(_run(["--gpus", "8", "--scaling", "weak"]).stdout)
    assert all(d["hbm_plan"] == one for d in weak) and [d["slice"] for d in weak] == [[40 * i, 40 * i + 40] for i in range(8)]
