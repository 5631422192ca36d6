"""bench.py as its own launcher (`python bench.py --gpus N`): the parent starts N fresh rank processes with the
torch.distributed environment and never touches HIP (tools/scripts/train.sh:19, tools/dist_test.py:85-129 do this
through torch.distributed.launch in the reference). CPU only: the ranks here run a gloo probe script."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PROBE = os.path.join(REPO, "tests", "_rank_probe.py")


def test_launcher_starts_n_ranks_with_the_distributed_environment(tmp_path):
    import bench
    rc = bench.launch_ranks(2, [str(tmp_path), "-1"], script=PROBE, timeout=120)
    assert rc == 0
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [s["rank"] for s in seen] == [0, 1] and [s["local_rank"] for s in seen] == [0, 1]
    assert all(s["world"] == 2 and s["sum"] == 1.0 and s["addr"] == "127.0.0.1" and s["launched"] == "1" for s in seen)


def test_launcher_reports_a_failing_rank_and_stops_the_others(tmp_path):
    import bench
    rc = bench.launch_ranks(2, [str(tmp_path), "1"], script=PROBE, timeout=60)
    assert rc != 0                                  # rank 1 exits 7 before the rendezvous; rank 0 is terminated


def test_parent_does_not_import_torch_before_launching():
    # the launcher decision is taken before torch (and with it any HIP runtime) is imported
    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; a = bench.parse(sys.argv[1:]); "
            "assert a.gpus == 2 and 'torch' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)")
    subprocess.check_call([sys.executable, "-c", code], cwd=REPO)


def test_gpus_must_match_world_size():
    # under an existing launcher (WORLD_SIZE set) a different --gpus is an error, not a silent 1-rank run
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--no-variants"], env=env,
                       capture_output=True, text=True, cwd=REPO)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (p.stderr + p.stdout)
