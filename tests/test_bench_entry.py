"""bench.py's entry points on the CPU (`--rehearsal`: no engine, a stand-in CPU step): the N > 1 paths the
driver uses -- a plain `python bench.py --gpus 2` that starts its own ranks as a child process, and the
same file launched under `torch.distributed.run` -- must rendezvous on 127.0.0.1 (gloo here), time between
barriers, reduce MAX over ranks and print exactly ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_plain_invocation_with_gpus_2_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearsal", "--steps", "20", "--warmup", "5",
                        "--envs", "512"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 5 and j["repeats"] == 100
    assert j["rehearsal"] is True and j["value"] is None and j["roofline"] is None
    assert j["scaling"] == "weak" and j["ms_per_step"] > 0


def test_driver_style_torchrun_launch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--rehearsal", "--steps", "7", "--warmup", "2", "--envs", "64"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 7


def test_a_failing_rank_fails_the_launcher():
    """A rank that dies AFTER the rendezvous (ACAS2D_BENCH_FAIL_RANK, a test hook inside the ranks) takes the job down:
    torchrun reports it, `python bench.py --gpus 2` hands the non-zero code back and prints no result line; the same
    command without the hook succeeds (test_plain_invocation_with_gpus_2_starts_its_own_ranks)."""
    env = dict(_env(), ACAS2D_BENCH_FAIL_RANK="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearsal", "--steps", "5", "--warmup", "1", "--envs", "64"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and not _json_lines(r.stdout)
    # an argument the parent itself rejects never starts any rank
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearsal", "--dtype", "f16"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 2


def test_pick_chunk_and_repeats():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.pick_chunk(20, 200) == 20 and bench.pick_chunk(2000, 200) == 200
    assert bench.pick_chunk(1000, 200) == 200 and bench.pick_chunk(7, 200) == 7
    assert bench.pick_chunk(1009, 200) == 200            # a prime: whole replays + an eager remainder
    a = bench.parse(["--steps", "20"])
    assert a.steps == 20 and a.repeats == 0 and a.gpus == 1


def test_counter_traffic_names_the_kernel_sources_it_was_taken_on():
    """profiles/traffic.json carries a digest of the kernel sources its rocprofv3 counter passes ran on; bench.py
    recomputes it and says in the line whether the timed sources are the same (`kernel_sources_match`)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rec = b.load_traffic(65536, 8, "f32", detail=True)
    assert rec is not None and len(rec["kernel_sources_sha256"]) == 64 and isinstance(rec["kernel_sources_match"], bool)
    assert len(b.kernel_sources_sha256()) == 64 and rec["build"] != "unlabelled"


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_real_ranks_on_one_gpu():
    """The N > 1 path on hardware, as far as one GPU goes: `python bench.py --gpus 2` starts two ranks that both use
    GPU 0 (--device 0, gloo for the barrier / MAX: RCCL wants one device per rank), each steps its own shard of
    65 536 envs with its own global env offset, rank 0 prints the one line.  (The 8-GPU curve is the driver's.)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--device", "0", "--steps", "100",
                        "--warmup", "20", "--no-rollout"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["envs_per_gpu"] == 65536
    # two ranks time-share one GPU: the whole-job rate is about the single-rank rate, never twice it
    assert 4e9 < j["value"] < 2.4e10 and j["config"]["episodes_finished"] > 1000
    assert "x2" in j["config"]["parallelism"] and j["roofline"]["frac"] > 0.15
