"""bench.py's N > 1 branch at world size 2 in the CPU suite: `python bench.py --gpus 2 --backend gloo
--extractor-factory bench_stub:make_extractor` goes through spawn_ranks -> torch.distributed.run -> two ranks -> gloo group, and runs the file's own step() /
pending[] / fence() / configs[4] / max-over-ranks code with CPU tensors; the slabs are written by tests/bench_stub.py (the oracle on host
memory - a REHEARSAL: the line says value = null; bench.py itself contains no stand-in extractor, it only imports the module the flag names).  Rank 0 checks frames of EVERY rank's gathered slab against the oracle on that rank's frames, and a wrong slab makes the
process exit non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra_env=None, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests")] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--extractor-factory", "bench_stub:make_extractor", "--batch", "3", "--steps", "3",
                        "--warmup", "1", *extra_args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    return r, (json.loads(lines[0]) if len(lines) == 1 else None)


def test_two_ranks_run_the_gather_branch_and_rank0_checks_what_arrived():
    r, j = run()
    assert r.returncode == 0, r.stderr[-4000:]
    assert j is not None and j["stub"] is True and j["value"] is None and j["n_gpus"] == 2 and j["backend"] == "gloo"
    v = j["verified"]
    assert v["bit_exact"] is True and v["ranks_checked"] == [0, 1] and v["frames_of_other_ranks"] == {"1": [1, 2]}
    assert v["frames"] == [0, 1, 2] and v["keypoints"] > 4000
    # the diagnostics a first real SCALE run needs: every rank's own step time, what the wait for the previous gather
    # cost each rank, and - on hardware - rank 0's kernels with the gather running (absent in the rehearsal, and said so)
    m = j["multi_gpu"]
    assert len(m["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in m["per_rank_ms_per_step"])
    assert max(m["per_rank_ms_per_step"]) == j["ms_per_step"]                      # the line's step time is the slowest rank's
    g = m["gather_exposed_ms"]
    assert len(g["host_blocked_ms_per_step"]) == 2 and len(g["stream_stalled_ms_per_step"]) == 2 and all(v >= 0 for v in g["host_blocked_ms_per_step"])
    assert m["rank0_kernel_ms_per_step"] is None and "rehearsal" in m["rank0_kernel_ms_note"]
    c4 = j["secondary"]["configs4_64_per_gpu"]       # BASELINE.json configs[4]'s code path (here 3 frames per rank), timed under the same rules
    assert c4["global_frames_per_step"] == 6 and c4["steps"] == 3 and c4["fps"] > 0 and "gathered to rank 0" in c4["note"]


def test_a_wrong_slab_from_another_rank_fails_the_run():
    """rank 1 corrupts one descriptor byte of the last timed step's slab (test switch): rank 0 must see it in what arrived and the
    launcher must exit non-zero with bit_exact = false in the line."""
    r, j = run({"ORBX_BENCH_TEST_CORRUPT_RANK": "1"})
    assert r.returncode != 0
    assert j is not None and j["verified"]["bit_exact"] is False and [1, 2] in j["verified"]["mismatching_frames"]


def test_no_gather_leaves_results_on_their_ranks():
    r, j = run(extra_args=("--no-gather",))
    assert r.returncode == 0, r.stderr[-4000:]
    assert j["verified"]["ranks_checked"] == [0] and j["verified"]["bit_exact"] is True
    assert j["multi_gpu"]["gather_exposed_ms"] is None and len(j["multi_gpu"]["per_rank_ms_per_step"]) == 2


def test_bench_py_names_the_oracle_only_in_its_checking_legs():
    """no line of bench.py may import the oracle outside the post-timing `verified` / `cpu_baseline` /
    `hd1080` legs; the stand-in extractor of the rehearsal lives in tests/bench_stub.py behind --extractor-factory."""
    src = open(os.path.join(ROOT, "bench.py")).read().splitlines()
    imports = [i for i, l in enumerate(src) if "import oracle_lib" in l]
    t0 = next(i for i, l in enumerate(src) if l.strip() == "t0 = time.perf_counter()")
    t1 = next(i for i, l in enumerate(src) if l.strip() == "elapsed = time.perf_counter() - t0")
    assert imports and all(i > t1 for i in imports), "an oracle import in front of / inside the timed region: %s" % imports
    assert t0 < t1
    for i in imports:
        assert "checker" in src[i], src[i]
    assert "StubExtractor" not in "\n".join(src) and "O_stub" not in "\n".join(src)
