"""`python bench.py --gpus N` as the driver types it: without a launcher in the environment the process must start its ranks itself,
as child processes, before it has imported torch or touched the GPU (BASELINE.json configs[4])."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_command_is_one_rank_per_gpu_on_localhost():
    import bench
    cmd = bench.spawn_command(["--gpus", "8", "--steps", "20", "--warmup", "3", "--spawn"], 8, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]       # own arguments passed through, --spawn dropped
    # a free port is chosen when none is given
    assert int(bench.spawn_command([], 2)[bench.spawn_command([], 2).index("--master-port") + 1]) > 0


def test_parent_starts_the_ranks_before_torch_is_imported():
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '4', '--steps', '2']\n"
            "import bench\n"
            "def fake(argv, gpus):\n"
            "    print('SPAWN', gpus, argv, 'torch' in sys.modules)\n"
            "    return 7\n"
            "bench.spawn_ranks = fake\n"
            "sys.exit(bench.main())\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, r.stderr            # the launcher's exit code is the parent's
    assert "SPAWN 4 ['--gpus', '4', '--steps', '2'] False" in r.stdout


def test_under_a_launcher_the_process_is_a_rank_not_a_parent():
    code = ("import sys; sys.argv = ['bench.py', '--gpus', '2']\n"
            "import bench\n"
            "bench.spawn_ranks = lambda a, g: (_ for _ in ()).throw(AssertionError('spawned under a launcher'))\n"
            "try:\n"
            "    bench.main()\n"
            "except SystemExit as e:\n"
            "    print('EXIT', e)\n")
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "spawned under a launcher" not in r.stderr + r.stdout
    assert "needs a GPU" in r.stdout or r.returncode == 0, r.stdout + r.stderr      # no GPU here: the rank refuses to run, loudly


@pytest.mark.gpu
def test_spawn_path_end_to_end_on_one_gpu():
    """The same code path as `--gpus 8` typed without a launcher, at the world size a one-GPU box has: parent -> torch.distributed.run
    -> rank -> RCCL process group -> HIP kernels -> gather -> one JSON line relayed by the parent; the line carries the oracle check
    of the slab it timed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ORBX_BENCH_CONFIGS4_AT_ANY_N"] = "1"            # the N > 1 line's secondary figure (BASELINE.json configs[4]) at the world size this box has
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and "RCCL gather" in j["config"]["parallelism"]
    assert j["verified"]["bit_exact"] is True and len(j["verified"]["frames"]) == 4
    c4 = j["secondary"]["configs4_64_per_gpu"]
    assert c4["frames_per_gpu_per_step"] == 64 and c4["fps"] > 0 and "gathered to rank 0" in c4["note"]
