"""bench.py's launch and rank plumbing on CPU: world sizes 2, 4 and 8 over gloo with the stub renderer (`--stub`).

What this pins ("make the multi-GPU entry unbreakable"):
  * `python bench.py --gpus 2` WITHOUT rank environment re-executes itself under torch.distributed.run and
    reports n_gpus == 2 — never a silent one-GPU run;
  * exactly one JSON line, from rank 0, with ranks_seen == [0, 1] and the communicator's own world size;
  * when the native pipeline fails on ONE rank, every rank agrees (all-reduce MIN) to take the torch path;
  * a WORLD_SIZE that contradicts --gpus is an error, and so is --gpus N on a node with fewer GPUs.
The renderer is a stub (it renders nothing): the traversal has no CPU implementation, and this test is about
the plumbing around it.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    return env


def _run(args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH, *args], cwd=ROOT, env=env or _env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


def _one_json_line(stdout: bytes) -> dict:
    lines = [ln for ln in stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, f"expected ONE line on stdout, got {len(lines)}: {lines}"
    return json.loads(lines[0])


@pytest.mark.timeout(400)
def test_gpus_2_without_rank_env_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--stub"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 2
    assert out["ranks_seen"] == [0, 1]
    assert out["rccl_world"] == 2          # what the (stub) communicator reports, min over ranks
    assert out["dist_path"] == "native"
    assert out["steps"] == 6 and out["warmup"] == 2
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "re-executing" in r.stderr.decode()
    # two timed legs in one invocation: north_star's literal one-gather-per-frame — `value` — and the batched one, `value_batched`
    assert set(out["legs"]) == {"batch1", "batch8"}
    assert out["value"] == out["legs"]["batch1"]["value"] and out["ms_per_step"] == pytest.approx(out["legs"]["batch1"]["ms_per_step"])
    assert out["value_batched"] == out["legs"]["batch8"]["value"]
    assert "ONE collective per frame" in out["config"]["parallelism"]
    # per-phase wall clock of the run and the budget the secondary leg has to fit
    assert {"startup", "count_rays", "native_probe", "contexts_and_root_share_tuning", "timed_legs", "secondary_leg", "secondary_cfg4_leg"} <= set(out["phase_seconds"])
    assert sum(out["phase_seconds"].values()) < out["wall_budget_s"] == 900.0
    assert out["legs"]["batch1"]["frames_per_collective"] == 1 and out["legs"]["batch8"]["frames_per_collective"] == 8
    # the root share comes from a warm-up auto-tune whose winner every rank agrees on (times are maxima over ranks); the stub's
    # frames are fastest at a share of 60 %
    for leg in ("batch1", "batch8"):
        tune = out["root_share_tuning"][leg]
        assert tune["tuned"] and set(tune["candidates_ms_per_frame"]) == {"60", "100"} and tune["root_share"] == 60
        assert out["legs"][leg]["root_share_percent"] == 60
        bd = out["legs"][leg]["breakdown"]
        assert [row["rank"] for row in bd["per_rank"]] == [0, 1] and bd["max_over_ranks"]["kernel_us_per_frame"] > 0
    # and the same pipeline on BASELINE's sharded configurations as secondary legs: configs[3] with the literal
    # one-gather-per-frame leg — its `value` — beside the batched one, the root share tuned once; configs[4], the path trace, literal, equal share
    _check_secondary(out)
    assert out["secondary"]["legs"]["batch1"]["root_share"]["root_share"] == 60


def _check_secondary(out):
    sec = out["secondary"]
    assert sec["workload"] == "cfg3_4k_1024c_b8" and set(sec["legs"]) == {"batch1", "batch8"}
    assert sec["value"] == sec["legs"]["batch1"]["value"] > 0 and sec["value_batched"] == sec["legs"]["batch8"]["value"] > 0
    assert sec["frames_per_collective"] == 1 and sec["legs"]["batch8"]["frames_per_collective"] == 8
    assert sec["legs"]["batch1"]["root_share"]["tuned"] and not sec["legs"]["batch8"]["root_share"]["tuned"]
    assert sec["legs"]["batch8"]["root_share"]["root_share"] == sec["legs"]["batch1"]["root_share"]["root_share"]
    assert [row["rank"] for row in sec["legs"]["batch1"]["breakdown"]["per_rank"]] == list(range(out["n_gpus"]))
    c4 = out["secondary_cfg4"]
    assert c4["workload"] == "cfg4_4k_2048c_b8_sparse" and set(c4["legs"]) == {"batch1"} and c4["value"] == c4["legs"]["batch1"]["value"] > 0
    assert c4["legs"]["batch1"]["root_share"] == {"tuned": False, "root_share": 100} and c4["value_batched"] is None


@pytest.mark.timeout(400)
def test_native_failure_on_one_rank_moves_every_rank_to_the_torch_path():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--stub", "--stub-fail-native-on", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1]
    assert out["dist_path"] == "torch"     # rank 0's native set-up worked, rank 1's did not: both fall back
    assert out["rccl_world"] is None
    assert "torch.distributed gather per frame" in out["config"]["parallelism"]
    assert set(out["legs"]) == {"torch"} and out["secondary"] is None and out["secondary_cfg4"] is None


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks_over_gloo(world):
    """the plumbing of the first 8-GPU contact at its real rank counts — both legs, the tuner's agreement across ranks, a
    breakdown row per rank, the secondary leg, the wall-clock phases."""
    sys.path.insert(0, ROOT)
    import bench
    r = _run(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--stub"], timeout=550)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == world and out["ranks_seen"] == list(range(world)) and out["rccl_world"] == world
    assert out["dist_path"] == "native" and set(out["legs"]) == {"batch1", "batch8"}
    assert out["value"] == out["legs"]["batch1"]["value"] and out["value_batched"] == out["legs"]["batch8"]["value"]
    cands = {str(c) for c in bench.root_share_candidates(world)}
    assert len(cands) == 3
    for leg in ("batch1", "batch8"):
        tune = out["root_share_tuning"][leg]
        # (every rank timed every candidate, the times are maxima over ranks, so the winner is one number for the whole job)
        assert tune["tuned"] and set(tune["candidates_ms_per_frame"]) == cands and str(tune["root_share"]) in cands
        assert out["legs"][leg]["root_share_percent"] == tune["root_share"]
        bd = out["legs"][leg]["breakdown"]
        assert [row["rank"] for row in bd["per_rank"]] == list(range(world))
    _check_secondary(out)
    # (first contact with an 8-GPU node is one shot: the whole run, both secondary legs included, inside the wall budget)
    assert sum(out["phase_seconds"].values()) < 900.0 and {"secondary_leg", "secondary_cfg4_leg"} <= set(out["phase_seconds"])


@pytest.mark.timeout(600)
def test_native_failure_on_one_of_eight_ranks_moves_every_rank_to_the_torch_path():
    r = _run(["--gpus", "8", "--steps", "4", "--warmup", "1", "--stub", "--stub-fail-native-on", "5"], timeout=550)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _one_json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["ranks_seen"] == list(range(8))
    assert out["dist_path"] == "torch" and out["rccl_world"] is None and set(out["legs"]) == {"torch"} and out["secondary"] is None


@pytest.mark.timeout(400)
def test_secondary_leg_is_skipped_when_the_wall_budget_is_spent():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--stub", "--wall-budget", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = _one_json_line(r.stdout)
    assert set(out["legs"]) == {"batch1", "batch8"} and "wall budget" in out["secondary"]["skipped"] and "wall budget" in out["secondary_cfg4"]["skipped"]


def test_world_size_contradicting_gpus_is_an_error():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub"], env=env)
    assert r.returncode != 0
    assert b"WORLD_SIZE=1" in r.stderr
    assert r.stdout.strip() == b""


def test_more_gpus_than_the_node_has_is_an_error():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 7), "--steps", "2", "--warmup", "0"])
    assert r.returncode != 0
    assert b"refusing to run on fewer" in r.stderr
    assert r.stdout.strip() == b""


def test_metric_string_follows_the_workload():
    sys.path.insert(0, ROOT)
    import bench
    from zig_vulkan_amd import workloads as W
    assert "1920x1080 on 512^3" in bench.metric_name(W.WORKLOADS[W.HEADLINE])
    assert "256x256 on 64^3" in bench.metric_name(W.WORKLOADS["cfg0_256x256_64c_b4"])
    assert "3840x2160 on 2048^3" in bench.metric_name(W.WORKLOADS["cfg4_4k_2048c_b8_sparse"])


def _import_bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


class _ProbeEnv:
    """The slice of bench.Env that native_probe uses, for one rank."""
    rank, world, local_rank = 0, 2, 0

    def bcast(self, obj):
        return obj

    def all_min_int(self, v):
        return v


@pytest.mark.parametrize("child, expect_ok, expect_text", [
    ("import sys; sys.exit(0)", True, "ok"),
    ("import sys; print('the shard of rank 1 never arrived', file=sys.stderr); sys.exit(3)", False, "never arrived"),
    ("import time; time.sleep(60)", False, "no answer within"),
])
def test_native_probe_turns_a_failing_or_hanging_child_into_the_torch_path(monkeypatch, child, expect_ok, expect_text):
    """N > 1: the native RCCL pipeline is first run by a child process of every rank (bench.native_probe).  A child that fails, or
    hangs in a collective, must cost a bounded wait and send every rank to the torch.distributed path — never hang the bench."""
    import subprocess
    import zig_vulkan_amd
    bench = _import_bench()
    monkeypatch.setattr(zig_vulkan_amd.VoxelRT, "dist_unique_id", staticmethod(lambda: bytes(128)))
    real_run = subprocess.run

    def fake_run(cmd, **kw):
        assert "--dist-probe" in cmd and "--probe-world" in cmd
        assert "RANK" not in kw["env"] and "WORLD_SIZE" not in kw["env"]   # the child must not join the parent's process group
        return real_run([sys.executable, "-c", child], **{k: v for k, v in kw.items() if k != "cwd"})

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    ok, report = bench.native_probe(_ProbeEnv(), timeout=2.0)
    assert ok is expect_ok and report["ok"] is expect_ok
    assert expect_text in report["this_rank"]
    assert report["seconds"] < 30


def test_profile_summariser_fails_loudly_without_the_per_phase_split(tmp_path):
    """VERDICT r05 (evidence regression): round 5's summaries lost the "launches by bench.py phase" section silently because the evidence
    script wrote the bench line where the summariser did not look.  With --require-phases the summariser now exits non-zero when it cannot
    print that section (no bench line found here: an empty directory), and says so in its output."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "summarize_prof.py"), str(tmp_path), "--require-phases"], capture_output=True, text=True)
    assert r.returncode == 2 and "NO bench line found" in r.stdout and "no phase split was printed" in r.stdout
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "summarize_prof.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "NO bench line found" in r.stdout      # (without the flag: a warning in the summary)
    sh = open(os.path.join(root, "tools", "evidence.sh")).read()
    assert "--bench-line $OUT/$W/bench.json --require-phases" in sh and "exit 3" in sh
