"""`python bench.py --gpus N` starts its own ranks: the parent must hand every child the rendezvous environment
torch.distributed expects, relay rank 0's single line, and fail when any rank fails.  A stub stands in for the ranks
(SS_BENCH_RANK_PROGRAM), so nothing here touches a GPU."""

import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent("""
    import json, os, sys
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
            "HSA_ENABLE_IPC_MODE_LEGACY")
    env = {k: os.environ.get(k) for k in keys}
    with open(os.path.join(os.environ["STUB_OUT"], "rank%s.json" % env["RANK"]), "w") as f:
        json.dump({"env": env, "argv": sys.argv[1:]}, f)
    print(json.dumps({"n_gpus": int(env["WORLD_SIZE"]), "from_rank": int(env["RANK"])}), flush=True)
    sys.exit(int(os.environ.get("STUB_FAIL_RANK", "-1")) == int(env["RANK"]) and 3 or 0)
""")


def _run(tmp_path, n, extra_env=None):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, SS_BENCH_RANK_PROGRAM=str(stub), STUB_OUT=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=120)


def test_parent_starts_one_rank_per_gpu_and_relays_rank0(tmp_path):
    r = _run(tmp_path, 3)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # only rank 0's line reaches the parent's stdout
    assert json.loads(lines[0]) == {"n_gpus": 3, "from_rank": 0}
    seen = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(3)]
    ports = {s["env"]["MASTER_PORT"] for s in seen}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for k, s in enumerate(seen):
        assert s["env"]["RANK"] == s["env"]["LOCAL_RANK"] == str(k)
        assert s["env"]["WORLD_SIZE"] == s["env"]["LOCAL_WORLD_SIZE"] == "3"
        assert s["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert s["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert s["argv"] == ["--gpus", "3", "--steps", "3", "--warmup", "1"]


def test_a_failing_rank_fails_the_parent(tmp_path):
    r = _run(tmp_path, 2, {"STUB_FAIL_RANK": "1"})
    assert r.returncode == 3
    assert "rank 1 exited with status 3" in r.stderr


def test_under_a_launcher_the_parent_does_not_spawn(tmp_path):
    # with WORLD_SIZE set (torch.distributed.run) bench.py is a rank itself: it must not start children.
    # Without a GPU it stops at the device check, after the launcher branch.
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    env = dict(os.environ, SS_BENCH_RANK_PROGRAM=str(stub), STUB_OUT=str(tmp_path), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert not (tmp_path / "rank0.json").exists()
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device" in r.stderr
