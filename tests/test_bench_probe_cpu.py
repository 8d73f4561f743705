"""bench.py's transport probe (N > 1: every transport is tried in a child process first) — the ORCHESTRATION, on the CPU.
There is no GPU here, so the probe child itself cannot succeed; what is checked is everything around it, under the
driver's own launcher: a fresh rendezvous port shipped over the group, the child started without the launcher's
TORCHELASTIC_* variables (with them it would wait for the agent's store and every probe would "hang" — found on the
first GPU run), its log captured, a child that fails reported as failed with its last lines, a child that never
answers killed at the deadline and reported as hung, and every rank agreeing on the verdict.  The success path and the
fall-back chain run on the GPU box (tests/test_gpu_comm.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def _run(extra_env, timeout, tmp_path):
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo", PROBE_VERDICT_DIR=str(tmp_path), **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    from launch import run_with_port
    r = run_with_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                    "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "probe_worker.py"), "capi"],
                      timeout=timeout, env=env, cwd=ROOT)
    # one file per rank: the ranks share the launcher's stdout pipe, and verdict lines printed there interleaved
    verdicts = [json.load(open(os.path.join(tmp_path, f))) for f in sorted(os.listdir(tmp_path)) if f.endswith(".json")]
    assert r.returncode == 0 and len(verdicts) == 2, (r.stdout[-2000:], r.stderr[-3000:])
    return sorted(verdicts, key=lambda v: v["rank"])


@pytest.mark.timeout(300)
def test_probe_child_that_fails_is_reported_by_every_rank(tmp_path):
    v = _run({"AH_BENCH_PROBE_TIMEOUT": "120"}, 280, tmp_path)
    assert [x["ok"] for x in v] == [False, False]
    assert v[0]["why"] == v[1]["why"], v  # the ranks agree on ONE verdict (all-gathered, first failing rank's words)
    assert "probe exited" in v[0]["why"] or "without PROBE_OK" in v[0]["why"], v
    assert "hung" not in v[0]["why"], v  # (a child that waits for the launcher's store would end here: TORCHELASTIC_* must be stripped)
    assert max(x["seconds"] for x in v) < 100, v


@pytest.mark.timeout(300)
def test_probe_child_that_never_answers_is_killed_at_the_deadline(tmp_path):
    v = _run({"AH_BENCH_PROBE_TIMEOUT": "4", "AH_BENCH_PROBE_TEST_SLEEP": "600"}, 280, tmp_path)
    assert [x["ok"] for x in v] == [False, False]
    assert "hung" in v[0]["why"] and "4 s" in v[0]["why"], v
    assert max(x["seconds"] for x in v) < 60, v
