"""bench.py's launch decision: `--gpus N` without a launcher environment starts the N-rank job as a child,
a mismatch between --gpus and WORLD_SIZE is an error in EVERY case, and a rank of a launched job runs in place.  Pure host logic:
no GPU, no process is started except a stand-in child for the relay."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_single_gpu_without_launcher_runs_in_place():
    assert bench.launch_plan(1, False, {}, [], 1) == ("rank", 0, 1, 0)


def test_n_gpus_without_launcher_spawns_the_job_with_the_same_arguments():
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    plan = bench.launch_plan(8, False, {}, argv, 8)
    assert plan[0] == "spawn"
    cmd = plan[1]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-len(argv) - 1] == os.path.abspath(bench.__file__) and cmd[-len(argv):] == argv


def test_force_ddp_on_one_gpu_takes_the_same_self_launch_path():
    plan = bench.launch_plan(1, True, {}, ["--gpus", "1", "--force-ddp"], 1)
    assert plan[0] == "spawn" and "--nproc-per-node=1" in plan[1]


def test_rank_of_a_launched_job_runs_in_place():
    env = {"RANK": "3", "WORLD_SIZE": "4", "LOCAL_RANK": "3"}
    assert bench.launch_plan(4, False, env, [], 8) == ("rank", 3, 4, 3)
    assert bench.launch_plan(1, True, {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, [], 1) == ("rank", 0, 1, 0)


@pytest.mark.parametrize("gpus,env", [(8, {"RANK": "0", "WORLD_SIZE": "1"}),        # the silent 1-GPU run of round 4
                                      (1, {"RANK": "0", "WORLD_SIZE": "2"}),
                                      (4, {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"}),
                                      (2, {"WORLD_SIZE": "4"}),
                                      (0, {})])
def test_world_size_mismatch_is_always_an_error(gpus, env):
    assert bench.launch_plan(gpus, False, env, [], 8)[0] == "error"


def test_more_ranks_than_devices_is_an_error():
    assert bench.launch_plan(8, False, {}, [], 1)[0] == "error"


def test_spawned_job_relays_one_json_line_and_the_exit_status(capfd):
    good = {"metric": "m", "value": 1.0, "n_gpus": 2}
    prog = ("import sys, json; print('NCCL version banner'); print(json.dumps(%r)); print('{not json}'); sys.stderr.write('warn\\n')" % (good,))
    assert bench.spawn_job([sys.executable, "-c", prog]) == 0
    out, err = capfd.readouterr()
    assert [json.loads(x) for x in out.strip().splitlines()] == [good]          # exactly one line on stdout
    assert "NCCL version banner" in err and "{not json}" in err
    assert bench.spawn_job([sys.executable, "-c", "import sys; sys.exit(7)"]) == 7
    assert bench.spawn_job([sys.executable, "-c", "print('no line')"]) != 0


def test_visible_gpu_count_reads_no_gpu_runtime(tmp_path):
    """The parent of a spawned job counts devices from the environment or the KFD topology files, never through torch / HIP."""
    assert bench.visible_gpu_count({"HIP_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert bench.visible_gpu_count({"ROCR_VISIBLE_DEVICES": ""}) == 0
    for i, simd in enumerate((0, 1024, 1024)):          # a CPU node and two GPU nodes
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\n")
    assert bench.visible_gpu_count({}, str(tmp_path)) == 2
    assert bench.visible_gpu_count({}, str(tmp_path / "missing")) is None
