"""GPU: BASELINE configs[4] on one GPU — one rank's shard (131,072 rows, H=20) of the 1,048,576-row sweep of
platform amplitude x phase (32 x 32 lattice) x 1,024 goal poses (mpc_amd/scenarios.py:make_sweep; the platform
family is the reference's tests/simulate_platform_motion.py:17,23,42-59, the arm pose mpc.srdf:16-31).

At this size the oracle cannot check every row, so the batch goes through size-independent properties — scenarios
are independent, so repeats, permutations, halves and sub-shards must reproduce the same rows bit for bit — plus
strided samples against the CPU oracle (evaluation) and against the host-only SLSQP driver, which is pinned call
by call to scipy's core (solves)."""
import numpy as np
import pytest
import torch

from helpers import assert_cost_close, assert_grad_close, oracle_for, params_for
from mpc_amd.batch import EvalResult, MpcContext, ScenarioBatch
from mpc_amd.params import MpcParams
from mpc_amd.scenarios import SWEEP_ROWS, DeviceKinematics, make_sweep, to_device

pytestmark = pytest.mark.gpu

PER_GPU = SWEEP_ROWS // 8          # 131,072


def _ctx(mp, N):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    return MpcContext(mp, max_scenarios=N, device=0)


def _outputs(ctx, N):
    e = lambda *shape: torch.empty(shape, dtype=torch.float64, device=ctx.device)
    return EvalResult(e(N), e(N, ctx.n), e(N, ctx.m), e(N, ctx.n, ctx.m))


def _fields(r):
    return (r.cost, r.grad, r.con, r.con_jac)


def test_sweep_shard_131072_eval_with_dense_jacobian():
    """Rank 5's rows of the linear-motion sweep: f, c, FD gradient and the dense FD dc/dx (31.7 GB) in one call."""
    N, first = PER_GPU, 5 * PER_GPU
    mp = params_for(20)
    ctx = _ctx(mp, N)
    hs = make_sweep(mp, DeviceKinematics(ctx), first, N, motion="linear", with_prev=True)
    sc, x = to_device(ctx, hs)
    a, b = _outputs(ctx, N), _outputs(ctx, N)
    ctx.eval_fd(sc, x, out=a)
    ctx.eval_fd(sc, x, out=b)                                    # (i) a repeat is bit-identical
    for u, v in zip(_fields(a), _fields(b)):
        assert torch.equal(u, v)
    h = N // 2                                                   # (ii) two half batches equal the full batch
    for t in _fields(b):
        t.fill_(float("nan"))
    for lo, hi in ((0, h), (h, N)):
        ctx.eval_fd(sc.slice(lo, hi), x[lo:hi], out=EvalResult(*(t[lo:hi] for t in _fields(b))))
    for u, v in zip(_fields(a), _fields(b)):
        assert torch.equal(u, v)
    perm = torch.randperm(N, device=ctx.device, generator=torch.Generator(device=ctx.device).manual_seed(5))
    scp = ctx.gather(sc, perm.to(torch.int32), ScenarioBatch.empty(ctx.device, N, 20, True))
    ctx.eval_fd(scp, x[perm].contiguous(), out=b)                # (iii) a permuted batch gives the permuted rows
    for u, v in zip(_fields(a)[:3], _fields(b)[:3]):
        assert torch.equal(u[perm], v)
    for lo in range(0, N, 16384):                                # the dense dc/dx in pieces: 4 GB at a time
        assert torch.equal(a.con_jac[perm[lo:lo + 16384]], b.con_jac[lo:lo + 16384])
    assert bool(torch.isfinite(a.cost).all()) and bool(torch.isfinite(a.grad).all())
    # (iv) a strided sample against the CPU oracle, dense dc/dx and its zero pattern included
    idx = np.arange(37, N, 1024)
    co = oracle_for(mp)
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    cost, grad, con, jac = co.eval_fd(sub)
    tidx = torch.from_numpy(idx).to(ctx.device)
    assert_cost_close(a.cost[tidx].cpu().numpy(), cost)
    assert_grad_close(a.grad[tidx].cpu().numpy(), grad)
    np.testing.assert_allclose(a.con[tidx].cpu().numpy(), con, rtol=1e-13, atol=1e-13)
    got = a.con_jac[tidx].cpu().numpy()
    np.testing.assert_allclose(got, jac, rtol=1e-9, atol=1e-10)
    assert np.array_equal(got == 0.0, jac == 0.0)
    # the shard is what any other split of the sweep makes for the same global rows
    part = make_sweep(mp, DeviceKinematics(ctx), first + 70000, 300, motion="linear", with_prev=True)
    scq, xq = to_device(ctx, part)
    rq = ctx.eval_fd(scq, xq)
    assert torch.equal(rq.cost, a.cost[70000:70300]) and torch.equal(rq.grad, a.grad[70000:70300])
    assert torch.equal(rq.con_jac, a.con_jac[70000:70300])


def test_sweep_shard_131072_resident_solves():
    """Rank 3's rows of the angular-motion sweep, one compute_control tick each on the device-resident SLSQP driver."""
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.native_solver import NativeBatchedSLSQP

    N, first = PER_GPU, 3 * PER_GPU
    mp = params_for(20)
    ctx = _ctx(mp, N)
    hs = make_sweep(mp, DeviceKinematics(ctx), first, N, motion="angular")
    keep = np.nonzero(hs.active)[0]                              # compute_control solves only these (ur_mpc.py:108)
    assert len(keep) > 0.9 * N
    sc_all, _ = to_device(ctx, hs)
    x0 = np.tile(hs.u_init, (1, mp.horizon))                     # the cold start of compute_control (ur_mpc.py:183-185)
    gather = lambda rows: ctx.gather(sc_all, torch.from_numpy(rows.astype(np.int32)).to(ctx.device),
                                     ScenarioBatch.empty(ctx.device, len(rows), 20, False))
    with ResidentSLSQP(ctx, capacity=N) as res:
        full = res.solve(gather(keep), x0[keep])
        assert full.n_host_fallback == 0
        assert full.success.all(), f"exit modes: {np.unique(full.status, return_counts=True)}"
        # a sub-shard made on its own ([first + 40000, first + 48192): what a 16-rank split would hand one rank)
        lo, cnt = 40000, 8192
        part = make_sweep(mp, DeviceKinematics(ctx), first + lo, cnt, motion="angular")
        pk = np.nonzero(part.active)[0]
        scp, _ = to_device(ctx, part)
        sub = res.solve(ctx.gather(scp, torch.from_numpy(pk.astype(np.int32)).to(ctx.device),
                                   ScenarioBatch.empty(ctx.device, len(pk), 20, False)),
                        np.tile(part.u_init, (1, mp.horizon))[pk])
    inside = (keep >= lo) & (keep < lo + cnt)
    np.testing.assert_array_equal(keep[inside] - lo, pk)
    np.testing.assert_array_equal(sub.x, full.x[inside])
    np.testing.assert_array_equal(sub.fun, full.fun[inside])
    np.testing.assert_array_equal(sub.nit, full.nit[inside])
    np.testing.assert_array_equal(sub.status, full.status[inside])
    sample = np.arange(11, len(keep), 1999)                      # 66 rows spread over the shard's 128 lattice cells
    with NativeBatchedSLSQP(ctx, threads=12, capacity=len(sample)) as host:
        want = host.solve(gather(keep[sample]), x0[keep[sample]])
    np.testing.assert_array_equal(full.x[sample], want.x)
    np.testing.assert_array_equal(full.nit[sample], want.nit)
    np.testing.assert_array_equal(full.status[sample], want.status)
    np.testing.assert_array_equal(full.fun[sample], want.fun)
    lim = np.tile(np.asarray(mp.v_lim), mp.horizon)
    assert np.all(np.abs(full.x) <= lim + 1e-12)
    # every goal pose of a lattice cell was solved: the cell's rows differ only in the goal
    cell = (first + keep) // 1024
    assert len(np.unique(cell)) >= 100


@pytest.mark.parametrize("kind", ["scipy_core", "resident", "resident_tick"])
def test_sweep_rows_match_the_reference(golden_dir, kind):
    """configs[4] pinned to the REAL reference: tests/golden/solve_H20_sweep.npz holds two compute_control ticks of the
    reference (oracle/gen_golden.py --sweep) for eleven rows from all over the 1,048,576-row sweep, both platform
    families.  The rows are made again here by make_sweep on the device kinematics (their inputs must be the
    fixture's), solved by the lock-step driver over scipy's core, by the device-resident solver, and by the resident
    tick (start points from mpc_prepare_tick, warm start kept on the device), and compared at the north star's 1e-6 rad."""
    import os

    from mpc_amd.batch import MpcContext
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.scenarios import DeviceKinematics, make_sweep
    from mpc_amd.solver import BatchedMPC

    path = os.path.join(golden_dir, "solve_H20_sweep.npz")
    if not os.path.exists(path):
        pytest.fail("solve_H20_sweep.npz missing: python -m oracle.gen_golden --sweep (where /root/reference exists)")
    g = np.load(path, allow_pickle=False)
    N = g["q0"].shape[0]
    assert N >= 10 and set(g["motion"]) == {"linear", "angular"} and g["index"].max() >= (1 << 20) - 1024
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, max_scenarios=N, device=0)
    kin = DeviceKinematics(ctx)
    for k in range(N):                                     # the generator of the workload gives the fixture's inputs
        hs = make_sweep(mp, kin, int(g["index"][k]), 1, motion=str(g["motion"][k]))
        np.testing.assert_allclose(hs.q0[0], g["q0"][k], rtol=0, atol=1e-12)
        np.testing.assert_allclose(hs.state0[0, 12:18], g["ee"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(hs.target[0], g["target"][k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(hs.base[0], g["base"][k], rtol=0, atol=1e-12)
    solver = None if kind == "scipy_core" else ResidentSLSQP(ctx, capacity=N)
    try:
        bm = BatchedMPC(ctx, solver=solver, prep="resident" if kind == "resident_tick" else "host")
        for tick in range(g["control"].shape[1]):          # tick 1: warm start, damping term, EMA
            u = bm.compute_control(g["q0"], g["ee"], g["target"], g["base"])
            assert bm.last.success.all()
            np.testing.assert_allclose(bm.previous_solution, g["solution"][:, tick], rtol=0, atol=1e-6)
            np.testing.assert_allclose(u, g["control"][:, tick], rtol=0, atol=1e-6)
    finally:
        if solver is not None:
            solver.close()
