"""CPU: `bench.py --gpus N` starts its own ranks (children, before any GPU call), the per-scenario solve
results are gathered in rank order, and one JSON line comes back.  Uses the no-GPU rehearsal mode (gloo,
synthetic rows that depend only on the global scenario index): this is about launching, sharding and the
gather, the arithmetic is covered by the gpu-marked tests."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, env=e)


def test_gpus_2_launches_two_ranks_and_gathers_in_rank_order():
    r = _run("--gpus", "2", "--rehearse-launch", "--scenarios", "37")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # stdout carries rank 0's line and nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "scenario-dp2"
    assert line["solves"]["gathered"]["rows"] == 2 * 37
    # the gathered table is the single-process table (the ranks assert it row for row; rank 0 reports its sum)
    want = sum(float(np.sin(np.arange(74) * (k + 1.0)).sum()) for k in range(9))
    assert abs(line["solves"]["gathered"]["checksum"] - want) < 1e-9
    assert line["solves"]["host_threads_per_gpu"] >= 1        # not torch.distributed.run's default of one thread
    one = _run("--gpus", "1", "--rehearse-launch", "--scenarios", "37")
    assert one.returncode == 0 and json.loads(one.stdout)["n_gpus"] == 1


def test_launch_refuses_a_mismatched_launcher():
    r = _run("--gpus", "2", "--rehearse-launch", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "launcher started 1 ranks" in r.stderr


def test_two_rank_rehearsal_gathers_sweep_rows():
    """configs[4]: rank r contributes rows [131072 r, 131072 r + S) of the 32x32 lattice x 1,024 goal poses sweep; the
    gathered table is what one process makes for those global rows (the ranks assert it row for row)."""
    sys.path.insert(0, ROOT)
    from mpc_amd.scenarios import SWEEP_ROWS, sweep_goal_table, sweep_index, sweep_lattice

    S = 1500                                                  # crosses a lattice-cell boundary (1,024 goals per cell)
    r = _run("--gpus", "2", "--rehearse-launch", "--workload", "sweep", "--scenarios", str(S), "--motion", "angular")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and "sweep" in line["config"]["workload"]
    assert line["solves"]["gathered"]["rows"] == 2 * S
    gidx = np.concatenate([k * (SWEEP_ROWS // 8) + np.arange(S) for k in range(2)])
    ia, ip, goal = sweep_index(gidx)
    A, phi = sweep_lattice("angular")
    want = A[ia].sum() + phi[ip].sum() + sweep_goal_table(20)[goal, :6].sum() + goal.sum()
    assert abs(line["solves"]["gathered"]["checksum"] - want) < 1e-6 * abs(want)
    assert ia.max() == 4 and set(np.unique(ip)) == {0, 1}      # rank 1 starts at lattice row 4 (131,072 / 1,024 / 32)


def test_a_failing_rank_fails_the_parent():
    r = _run("--gpus", "2", "--rehearse-launch", "--scenarios", "5", env={"MPC_BENCH_REHEARSE_FAIL_RANK": "1"})
    assert r.returncode != 0 and not r.stdout.strip()


def test_solve_result_rows_layout():
    sys.path.insert(0, ROOT)
    import bench

    u = np.arange(24.0).reshape(4, 6)
    last = SimpleNamespace(fun=np.array([1.5, 2.5]), nit=np.array([7, 9]), status=np.array([0, 8]))
    rows = bench.solve_result_rows(u, last, np.array([0, 3]))
    assert rows.shape == (4, 9)
    np.testing.assert_array_equal(rows[:, 1:7], u)
    np.testing.assert_array_equal(rows[:, 0], [1.5, 0, 0, 2.5])
    np.testing.assert_array_equal(rows[:, 7], [7, 0, 0, 9])
    np.testing.assert_array_equal(rows[:, 8], [0, -1, -1, 8])     # -1: early exit (ur_mpc.py:108)
    np.testing.assert_array_equal(bench.solve_result_rows(u, None, np.zeros(0, int))[:, 8], -1)
