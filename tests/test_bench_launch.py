"""bench.py's multi-GPU launch path on CPU: `--gpus N` without a launcher must start N fresh rank processes,
rendezvous them, cut the batch with shard.shard_range, time with barriers and a max over ranks, and print ONE JSON
line from rank 0 with n_gpus = N.  `--dry-backend gloo` is the seam: same plumbing, nothing traced, no GPU touched
(the traversal path itself has no CPU form).  Under an external launcher (torchrun sets WORLD_SIZE) the script must
not spawn again, and a WORLD_SIZE that disagrees with --gpus is an error, not a silent 1-GPU run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def json_lines(p):
    return [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = run(["--gpus", "2", "--dry-backend", "gloo", "--steps", "3", "--warmup", "1", "--rays", "1001", "--scaling", "strong"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p)
    assert len(lines) == 1
    assert len(p.stdout.strip().splitlines()) == 1          # nothing else on stdout: gloo's own announcements go to stderr
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["dry"] is True and j["scaling"] == "strong"
    # two ranks answered, and their slices tile the batch: contiguous, disjoint, sizes differing by at most one
    assert [s[0] for s in j["slices"]] == [0, 1]
    assert j["slices"][0][1] == 0 and j["slices"][0][2] == j["slices"][1][1] and j["slices"][1][2] == 1001
    assert abs((j["slices"][0][2] - j["slices"][0][1]) - (j["slices"][1][2] - j["slices"][1][1])) <= 1
    # the radiance combine (all-reduce) saw both ranks: 1 + 2, and 3 + 3 samples
    assert j["combined_radiance"] == [3.0] * 4 and j["combined_spp"] == 6
    # max over ranks: rank 1 sleeps 2 ms per step
    assert j["ms_per_step"] >= 2.0


def test_three_ranks_uneven_slices():
    p = run(["--gpus", "3", "--dry-backend", "gloo", "--steps", "1", "--rays", "10"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json_lines(p)[0]
    assert j["n_gpus"] == 3 and j["scaling"] == "weak"
    assert [tuple(s) for s in j["slices"]] == [(0, 0, 4), (1, 4, 7), (2, 7, 10)]


def test_one_gpu_does_not_spawn():
    p = run(["--gpus", "1", "--dry-backend", "gloo", "--steps", "2", "--rays", "64"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json_lines(p)[0]
    assert j["n_gpus"] == 1 and [tuple(s) for s in j["slices"]] == [(0, 0, 64)] and j["combined_spp"] == 3


def test_under_a_launcher_it_joins_instead_of_spawning():
    """What the driver does: N processes with RANK / WORLD_SIZE already set, each given --gpus N."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(2):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-backend", "gloo", "--steps", "1", "--rays", "100"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    l0 = [x for x in outs[0][0].splitlines() if x.startswith("{")]
    l1 = [x for x in outs[1][0].splitlines() if x.startswith("{")]
    assert len(l0) == 1 and len(l1) == 0                       # rank 0 prints, rank 1 is silent
    assert json.loads(l0[0])["n_gpus"] == 2


def test_world_size_must_match_gpus():
    p = run(["--gpus", "1", "--dry-backend", "gloo"], env={"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in (p.stderr + p.stdout)


def test_a_failing_rank_fails_the_launcher():
    p2 = run(["--gpus", "2", "--dry-backend", "gloo", "--workload", "no-such-workload"])
    assert p2.returncode != 0


def test_a_rank_that_dies_mid_run_stops_the_others_within_seconds():
    """Rank 1 exits after the timed steps while rank 0 waits in the barrier: without supervision rank 0 would sit there
    until the process-group timeout (minutes).  The launcher must stop it and exit non-zero with every return code."""
    import time
    t0 = time.time()
    p = run(["--gpus", "2", "--dry-backend", "gloo", "--steps", "1", "--rays", "100"],
            env={"BENCH_DRY_DIE_RANK": "1", "BENCH_DRY_DIE_CODE": "7"}, timeout=120)
    dt = time.time() - t0
    assert p.returncode != 0
    assert "rank 1 exited with code 7" in p.stderr and "rank exit codes" in p.stderr, p.stderr[-2000:]
    assert dt < 60, "the launcher waited %.0f s for a rank that was never going to finish" % dt
    assert json_lines(p) == []                                    # no result line from a run that lost a rank


def test_the_launcher_gives_up_at_its_own_deadline():
    """All ranks alive but never finishing (rank 1 dies SILENTLY is covered above; here rank 0 is the one that dies, with
    code 0 masked as success being impossible -- so instead: a deadline of one second on a run that needs longer)."""
    p = run(["--gpus", "2", "--dry-backend", "gloo", "--steps", "4000", "--rays", "100"], env={"BENCH_LAUNCH_TIMEOUT": "1"}, timeout=120)
    assert p.returncode != 0 and "BENCH_LAUNCH_TIMEOUT reached" in p.stderr, p.stderr[-2000:]
