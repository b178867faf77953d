"""bench.py's own launcher (VERDICT r2 next #1a): `python bench.py --gpus N` without a torch.distributed environment must
start its N rank processes itself — as a CHILD, decided before torch or the GPU is touched — relay rank 0's single JSON line
and propagate failure.  Runs here without a GPU: ALAC_BENCH_STUB makes the ranks run the launcher contract on gloo
(rendezvous on 127.0.0.1, barrier, max over ranks, rank 0 prints)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(env_extra, *args):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, env=env,
                          timeout=300)


def test_plain_command_starts_two_ranks_and_relays_one_line():
    p = run({"ALAC_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "5", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE line on stdout: the ranks' chatter went to stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["stub"] is True
    assert out["value"] == 2.0  # max over the ranks of (1 + rank): the all-reduce ran over both
    assert "chatter" in p.stderr


def test_failing_rank_fails_the_command():
    p = run({"ALAC_BENCH_STUB": "fail"}, "--gpus", "2")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]


def test_launch_decision_is_made_before_torch_is_imported():
    """the parent must not import torch (and so cannot initialise the GPU) before it starts the ranks"""
    code = ("import sys, json; sys.argv = ['bench.py', '--gpus', '2']; import bench; "
            "seen = {}; "
            "bench.self_launch = lambda a, argv, run=None: (seen.update(torch='torch' in sys.modules, argv=argv), 0)[1]; "
            "rc = bench.main(); print(json.dumps(dict(rc=rc, **seen)))")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out == {"rc": 0, "torch": False, "argv": ["--gpus", "2"]}


def test_launch_command_shape():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "20"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
