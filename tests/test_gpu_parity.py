"""GPU: the HIP path (through the C-ABI) against the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_cost_close, assert_grad_close, host_scenarios, params_for
from mpc_amd import _ffi
from mpc_amd.batch import MpcContext
from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

pytestmark = pytest.mark.gpu

ALGOS = [_ffi.ALGO_ROLLOUT, _ffi.ALGO_PREFIX]


def _ctx(mp, N):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    return MpcContext(mp, max_scenarios=N, device=0)


def _supported(mp):
    try:
        ctx = MpcContext(mp, 1, 0)
        ctx.close()
        return True
    except _ffi.MpcError as e:
        if e.code == -3:
            return False
        raise


def _compare(res, cost, grad, con, jac, tight_con=True):
    assert_cost_close(res.cost.cpu().numpy(), cost)
    assert_grad_close(res.grad.cpu().numpy(), grad)
    np.testing.assert_allclose(res.con.cpu().numpy(), con, rtol=1e-13, atol=1e-13)
    if jac is not None:
        # entries are +-dt-sized FD quotients of joint sums; everything else is exactly zero
        got = res.con_jac.cpu().numpy()
        np.testing.assert_allclose(got, jac, rtol=1e-9, atol=1e-10)
        assert np.array_equal(got == 0.0, jac == 0.0)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("H,motion,prev", [(1, "linear", False), (1, "angular", True), (3, "linear", True),
                                            (3, "angular", False), (20, "linear", False), (20, "angular", True)])
def test_eval_fd_matches_oracle(algo, H, motion, prev):
    mp = params_for(H, algo=algo)
    if not _supported(mp):
        pytest.skip("algo not available for these parameters")
    N = 96
    co, hs = host_scenarios(mp, N, motion=motion, cfg_id=11, with_prev=prev, near_bounds=True)
    cost, grad, con, jac = co.eval_fd(hs.as_oracle_dict())
    ctx = _ctx(mp, N)
    sc, x = to_device(ctx, hs)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    _compare(res, cost, grad, con, jac)
    # f and c only (SLSQP mode 1) agree with the fd call's f and c bit for bit
    res1 = ctx.eval(sc, x)
    c1, con1 = co.eval(hs.as_oracle_dict())
    assert_cost_close(res1.cost.cpu().numpy(), c1)
    np.testing.assert_allclose(res1.con.cpu().numpy(), con1, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("algo", ALGOS)
def test_long_horizon_h100(algo):
    mp = params_for(100, algo=algo)
    if not _supported(mp):
        pytest.skip("algo not available for these parameters")
    N = 6
    co, hs = host_scenarios(mp, N, motion="linear", cfg_id=4, with_prev=True, near_bounds=True)
    cost, grad, con, jac = co.eval_fd(hs.as_oracle_dict())
    ctx = _ctx(mp, N)
    sc, x = to_device(ctx, hs)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    _compare(res, cost, grad, con, jac)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("H", [1, 3, 20, 100])
def test_against_reference_golden(golden_dir, algo, H):
    path = os.path.join(golden_dir, f"eval_H{H}.npz")
    if not os.path.exists(path):
        pytest.skip("golden not generated")
    g = np.load(path, allow_pickle=False)
    mp = params_for(H, algo=algo)
    if not _supported(mp):
        pytest.skip("algo not available for these parameters")
    N = g["x"].shape[0]
    ctx = _ctx(mp, N)
    from mpc_amd.batch import ScenarioBatch

    sc = ScenarioBatch.from_numpy(ctx.device, g["state0"], g["target"], g["base_traj"], g["Jc"], g["u_prev"],
                                  g["with_prev"])
    x = torch.as_tensor(g["x"], dtype=torch.float64).to(ctx.device)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    jac = np.transpose(g["jac"], (0, 2, 1)) if "jac" in g.files else None
    _compare(res, g["f"], g["grad"], g["c"], jac)


@pytest.mark.parametrize("algo", ALGOS)
def test_against_reference_golden_h100_angular(golden_dir, algo):
    """configs[3]'s horizon with the revolute platform: f, c, grad f and the full dc/dx of the REAL reference at H=100
    (tests/golden/eval_H100_angular.npz, oracle/gen_golden.py --h100), zero pattern included."""
    path = os.path.join(golden_dir, "eval_H100_angular.npz")
    if not os.path.exists(path):
        pytest.fail("eval_H100_angular.npz missing: python -m oracle.gen_golden --h100 (where /root/reference exists)")
    g = np.load(path, allow_pickle=False)
    mp = params_for(100, algo=algo)
    if not _supported(mp):
        pytest.skip("algo not available for these parameters")
    N = g["x"].shape[0]
    ctx = _ctx(mp, N)
    from mpc_amd.batch import ScenarioBatch

    sc = ScenarioBatch.from_numpy(ctx.device, g["state0"], g["target"], g["base_traj"], g["Jc"], g["u_prev"], g["with_prev"])
    x = torch.as_tensor(g["x"], dtype=torch.float64).to(ctx.device)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    assert (g["motion"] == "angular").all() and g["jac"].shape == (N, mp.m, mp.n)
    _compare(res, g["f"], g["grad"], g["c"], np.transpose(g["jac"], (0, 2, 1)))


def test_full_size_h100_16384():
    """BASELINE configs[3] at full size: 16,384 scenarios at H=100 with the dense dc/dx (95 GB of it; 288 GB of HBM).
    (i) the pipelined two-stream route over the whole batch equals two half batches bit for bit (different chunking
    of the workspace, different overlap), (ii) the structured form expands to the same matrix, (iii) a strided sample
    equals the CPU oracle including dc/dx and its zero pattern, (iv) state 0 and the upper triangle are exact zeros."""
    N, H = 16384, 100
    mp = params_for(H)
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=4, with_prev=True)
    sc, x = to_device(ctx, hs)
    full = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    assert full.con_jac.numel() * 8 > 95e9
    h = N // 2
    half_jac = torch.empty((h, mp.n, mp.m), dtype=torch.float64, device=ctx.device)
    from mpc_amd.batch import EvalResult
    for lo in (0, h):
        part = ctx.eval_fd(sc.slice(lo, lo + h), x[lo:lo + h],
                           out=EvalResult(torch.empty(h, dtype=torch.float64, device=ctx.device), None, None, half_jac))
        torch.cuda.synchronize()
        assert torch.equal(part.cost, full.cost[lo:lo + h]) and torch.equal(part.grad, full.grad[lo:lo + h])
        assert torch.equal(part.con, full.con[lo:lo + h])
        for a in range(0, h, 1024):                       # compare in slices: keeps the temporaries small
            assert torch.equal(half_jac[a:a + 1024], full.con_jac[lo + a:lo + a + 1024])
    z = ctx.eval_fd_nz(sc.slice(0, 2048), x[:2048])
    ctx.expand_jacobian(z.con_jac_nz, out=half_jac[:2048])
    torch.cuda.synchronize()
    assert torch.equal(half_jac[:2048], full.con_jac[:2048]) and torch.equal(z.grad, full.grad[:2048])
    del half_jac
    idx = np.arange(5, N, 2731)                           # 6 scenarios
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    c_o, g_o, con_o, jac_o = co.eval_fd(sub)
    tidx = torch.as_tensor(idx, device=ctx.device)
    assert_cost_close(full.cost[tidx].cpu().numpy(), c_o)
    assert_grad_close(full.grad[tidx].cpu().numpy(), g_o)
    np.testing.assert_allclose(full.con[tidx].cpu().numpy(), con_o, rtol=1e-13, atol=1e-13)
    got = full.con_jac[tidx].cpu().numpy()
    np.testing.assert_allclose(got, jac_o, rtol=1e-9, atol=1e-10)
    assert np.array_equal(got == 0.0, jac_o == 0.0)
    J = full.con_jac[:256].view(256, H, 6, H + 1, 12)     # [s][j][a][k][row in state]
    assert float(J[:, :, :, 0, :].abs().max()) == 0.0
    upper = torch.arange(H + 1, device=ctx.device)[None, :] <= torch.arange(H, device=ctx.device)[:, None]
    assert float(J.permute(0, 2, 4, 1, 3)[..., upper].abs().max()) == 0.0


def test_device_kinematics_match_oracle():
    mp = params_for(20)
    co, hs = host_scenarios(mp, 64, motion="angular", cfg_id=12)
    ctx = _ctx(mp, 64)
    q = torch.as_tensor(hs.q0).to(ctx.device)
    np.testing.assert_allclose(ctx.jacobian(q).cpu().numpy(), co.jacobian(hs.q0), rtol=0, atol=1e-14)
    pos, rpy = co.fk(hs.q0)
    pose = ctx.fk(q).cpu().numpy()
    np.testing.assert_allclose(pose[:, :3], pos, rtol=0, atol=1e-14)
    np.testing.assert_allclose(pose[:, 3:], rpy, rtol=0, atol=1e-13)
    bt = ctx.predict_base_dynamic(torch.as_tensor(hs.base).to(ctx.device)).cpu().numpy()
    np.testing.assert_allclose(bt, co.predict_base_dynamic(hs.base), rtol=1e-14, atol=1e-15)
    # the whole scenario family built on the device equals the one built with the oracle
    hd = make_scenarios(mp, DeviceKinematics(ctx), 64, motion="angular", cfg_id=12)
    np.testing.assert_allclose(hd.state0, hs.state0, rtol=0, atol=1e-13)
    np.testing.assert_allclose(hd.x, hs.x, rtol=0, atol=1e-11)


def test_algos_agree_at_scale():
    """Size-independent property at the benchmark's shape: the two kernels agree with each other on
    4,096 scenarios (the oracle checks a strided sample of the same batch)."""
    N = 4096
    mp1 = params_for(20, algo=_ffi.ALGO_ROLLOUT)
    co, hs = host_scenarios(mp1, N, motion="linear", cfg_id=2)
    ctx1 = _ctx(mp1, N)
    sc, x = to_device(ctx1, hs)
    r1 = ctx1.eval_fd(sc, x)
    torch.cuda.synchronize()
    idx = np.arange(0, N, 64)
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    cost, grad, con, jac = co.eval_fd(sub)
    assert_cost_close(r1.cost.cpu().numpy()[idx], cost)
    assert_grad_close(r1.grad.cpu().numpy()[idx], grad)
    mp2 = params_for(20, algo=_ffi.ALGO_PREFIX)
    if not _supported(mp2):
        pytest.skip("prefix kernel not available")
    ctx2 = _ctx(mp2, N)
    r2 = ctx2.eval_fd(sc, x)
    torch.cuda.synchronize()
    assert_cost_close(r2.cost.cpu().numpy(), r1.cost.cpu().numpy())
    assert_grad_close(r2.grad.cpu().numpy(), r1.grad.cpu().numpy())
    np.testing.assert_allclose(r2.con_jac.cpu().numpy(), r1.con_jac.cpu().numpy(), rtol=1e-9, atol=1e-10)


def test_empty_batch_and_bad_arguments():
    mp = params_for(3)
    ctx = _ctx(mp, 8)
    co, hs = host_scenarios(mp, 4)
    sc, x = to_device(ctx, hs)
    res = ctx.eval_fd(sc.slice(0, 0), x[0:0])          # empty batch is a no-op
    assert res.cost.numel() == 0
    with pytest.raises(ValueError):
        ctx.eval_fd(sc, x[:, :-1].contiguous())        # wrong n
    with pytest.raises(ValueError):
        ctx.eval_fd(sc, x.float())                     # wrong dtype
    with pytest.raises(ValueError):
        ctx.eval_fd(sc, x.cpu())                       # host tensor
    big, xb = to_device(ctx, host_scenarios(mp, 9)[1])
    with pytest.raises(ValueError):
        ctx.eval_fd(big, xb)                           # more scenarios than the context was sized for


def test_full_size_properties_65536_angular():
    """BASELINE configs[2] shape (65,536 angular-motion scenarios, H=20) through size-independent
    properties: scenarios are independent, so (i) a permuted batch gives the permuted outputs bit for
    bit, (ii) two half batches equal the full batch, (iii) a repeat launch is bit-identical; a strided
    sample is checked against the CPU oracle."""
    N = 65536
    mp = params_for(20)
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=3, with_prev=True)
    sc, x = to_device(ctx, hs)
    r = ctx.eval_fd(sc, x, want_jac=False)
    torch.cuda.synchronize()
    cost, grad, con = r.cost.clone(), r.grad.clone(), r.con.clone()
    r2 = ctx.eval_fd(sc, x, want_jac=False)
    assert torch.equal(r2.cost, cost) and torch.equal(r2.grad, grad) and torch.equal(r2.con, con)
    perm = torch.randperm(N, device=ctx.device, generator=torch.Generator(device=ctx.device).manual_seed(1))
    from mpc_amd.solver import _gather
    scp = _gather(sc, perm.cpu().numpy(), ctx.device)
    from mpc_amd.batch import ScenarioBatch
    scg = ctx.gather(sc, perm.to(torch.int32), ScenarioBatch.empty(ctx.device, N, 20, True))     # library gather
    for a, b in ((scg.state0, scp.state0), (scg.target, scp.target), (scg.base_traj, scp.base_traj), (scg.Jc, scp.Jc),
                 (scg.u_prev, scp.u_prev), (scg.has_prev, scp.has_prev)):
        assert torch.equal(a, b)
    bad = ctx.gather(sc, torch.tensor([5, -1, N, 7], dtype=torch.int32, device=ctx.device),
                     ScenarioBatch.empty(ctx.device, 4, 20, True))
    assert torch.equal(bad.state0[0], sc.state0[5]) and torch.equal(bad.base_traj[3], sc.base_traj[7])
    assert bool(torch.isnan(bad.state0[1:3]).all()) and bool(torch.isnan(bad.base_traj[1:3]).all())
    rp = ctx.eval_fd(scp, x[perm].contiguous(), want_jac=False)
    assert torch.equal(rp.cost, cost[perm]) and torch.equal(rp.grad, grad[perm]) and torch.equal(rp.con, con[perm])
    h = N // 2
    ra = ctx.eval_fd(sc.slice(0, h), x[:h], want_jac=False)
    rb = ctx.eval_fd(sc.slice(h, N), x[h:], want_jac=False)
    assert torch.equal(torch.cat([ra.cost, rb.cost]), cost) and torch.equal(torch.cat([ra.grad, rb.grad]), grad)
    idx = np.arange(0, N, 1024)
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    c_o, g_o, con_o, _ = co.eval_fd(sub, want_jac=False)
    assert_cost_close(cost.cpu().numpy()[idx], c_o)
    assert_grad_close(grad.cpu().numpy()[idx], g_o)
    np.testing.assert_allclose(con.cpu().numpy()[idx], con_o, rtol=1e-13, atol=1e-13)
    # dense Jacobian on a 4,096-scenario slice: exact zero pattern, and the non-zeros reproduce the
    # constraint's own finite difference (-sign(u) on the velocity rows, +-dt-sized on the position rows)
    rj = ctx.eval_fd(sc.slice(0, 4096), x[:4096])
    J = rj.con_jac.view(4096, 20, 6, 21, 12)            # [s][j][a][k][row in state]
    assert float(J[:, :, :, 0, :].abs().max()) == 0.0   # state 0 does not depend on x
    kidx = torch.arange(21, device=ctx.device)
    jidx = torch.arange(20, device=ctx.device)
    upper = (kidx[None, :] <= jidx[:, None])            # states k <= j are untouched by u_j
    assert float(J.permute(0, 2, 4, 1, 3)[..., upper].abs().max()) == 0.0
    vel = torch.stack([J[:, j, :, j + 1, 6:] for j in range(20)], dim=1)   # [s][j][a][6]
    diag = torch.diagonal(vel, dim1=2, dim2=3)
    u = x[:4096].view(4096, 20, 6)
    away = u.abs() > 2e-3                               # the step does not straddle |u| = 0 there
    assert torch.allclose(diag[away], -torch.sign(u)[away], atol=1e-9)


def test_h100_batch_in_chunks_equals_one_pass():
    """Long-horizon path: the workspace-chunked two-kernel route is independent of the batch split."""
    mp = params_for(100)
    N = 48
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="linear", cfg_id=4)
    sc, x = to_device(ctx, hs)
    full = ctx.eval_fd(sc, x)
    parts = [ctx.eval_fd(sc.slice(a, a + 16), x[a:a + 16]) for a in (0, 16, 32)]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([p.con_jac for p in parts]), full.con_jac)
    assert torch.equal(torch.cat([p.grad for p in parts]), full.grad)
    idx = np.array([0, 47])
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    c_o, g_o, con_o, jac_o = co.eval_fd(sub)
    assert_cost_close(full.cost.cpu().numpy()[idx], c_o)
    assert_grad_close(full.grad.cpu().numpy()[idx], g_o)
    np.testing.assert_allclose(full.con_jac.cpu().numpy()[idx], jac_o, rtol=1e-9, atol=1e-10)


def test_h100_two_stream_pipeline_equals_plain_chunks():
    """Batches of >= 2,048 long-horizon scenarios run the expansion of chunk i on a second stream while
    chunk i+1 rolls out: same bits as 1,024-scenario calls (which take the plain one-stream route),
    also when two calls follow each other without a sync (the workspace halves are reused in order)."""
    mp = params_for(100)
    N = 4096
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=9)
    sc, x = to_device(ctx, hs)
    full = ctx.eval_fd(sc, x)
    x2 = (x * 0.5).contiguous()
    again = ctx.eval_fd(sc, x2)            # no sync in between
    again2 = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    assert torch.equal(again2.con_jac, full.con_jac) and torch.equal(again2.grad, full.grad)
    for a in range(0, N, 1024):
        p = ctx.eval_fd(sc.slice(a, a + 1024), x[a:a + 1024])
        p2 = ctx.eval_fd(sc.slice(a, a + 1024), x2[a:a + 1024])
        torch.cuda.synchronize()
        assert torch.equal(p.con_jac, full.con_jac[a:a + 1024]) and torch.equal(p.grad, full.grad[a:a + 1024])
        assert torch.equal(p.cost, full.cost[a:a + 1024]) and torch.equal(p.con, full.con[a:a + 1024])
        assert torch.equal(p2.con_jac, again.con_jac[a:a + 1024]) and torch.equal(p2.cost, again.cost[a:a + 1024])
    idx = np.array([3, 2047, 2048, 4095])
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    c_o, g_o, con_o, jac_o = co.eval_fd(sub)
    assert_cost_close(full.cost.cpu().numpy()[idx], c_o)
    assert_grad_close(full.grad.cpu().numpy()[idx], g_o)
    np.testing.assert_allclose(full.con_jac[torch.as_tensor(idx, device=ctx.device)].cpu().numpy(), jac_o, rtol=1e-9, atol=1e-10)


def test_odd_long_horizon_through_the_two_stream_pipeline():
    """An ODD long horizon (H = 43: the middle perturbation has no partner, and the workspace's
    iteration-major layout has unused slots) at a batch size that takes the two-stream pipeline: same bits as
    calls small enough for the one-stream route, which the oracle checks."""
    mp = params_for(43)
    N = 2304
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=12, near_bounds=True)
    sc, x = to_device(ctx, hs)
    full = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    for a in range(0, N, 768):
        p = ctx.eval_fd(sc.slice(a, a + 768), x[a:a + 768])
        torch.cuda.synchronize()
        assert torch.equal(p.con_jac, full.con_jac[a:a + 768]) and torch.equal(p.grad, full.grad[a:a + 768])
        assert torch.equal(p.cost, full.cost[a:a + 768]) and torch.equal(p.con, full.con[a:a + 768])
    idx = np.array([0, 1151, 1152, 2303])
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    c_o, g_o, con_o, jac_o = co.eval_fd(sub)
    assert_cost_close(full.cost.cpu().numpy()[idx], c_o)
    assert_grad_close(full.grad.cpu().numpy()[idx], g_o)
    np.testing.assert_allclose(full.con_jac[torch.as_tensor(idx, device=ctx.device)].cpu().numpy(), jac_o, rtol=1e-9, atol=1e-10)
    z = ctx.eval_fd_nz(sc, x)
    assert torch.equal(ctx.expand_jacobian(z.con_jac_nz), full.con_jac)


@pytest.mark.parametrize("H,N", [(1, 37), (3, 29), (20, 512), (100, 24)])
def test_structured_jacobian_equals_dense(H, N):
    """mpc_eval_fd_nz: same f, grad f, c as mpc_eval_fd, and its structured dc/dx expands (on the device with
    mpc_expand_jacobian, on the host with slsqp_state.nz_scatter_index) to exactly mpc_eval_fd's dense matrix."""
    from mpc_amd.slsqp_state import nz_count, nz_scatter_index

    mp = params_for(H)
    ctx = _ctx(mp, N)
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=11, with_prev=True, near_bounds=True)
    sc, x = to_device(ctx, hs)
    dense = ctx.eval_fd(sc, x)
    z = ctx.eval_fd_nz(sc, x)
    torch.cuda.synchronize()
    assert ctx.nz_count == nz_count(H) and tuple(z.con_jac_nz.shape) == (N, nz_count(H))
    assert torch.equal(z.cost, dense.cost) and torch.equal(z.grad, dense.grad) and torch.equal(z.con, dense.con)
    assert torch.equal(ctx.expand_jacobian(z.con_jac_nz), dense.con_jac)
    idx = nz_scatter_index(H)
    assert len(np.unique(idx)) == len(idx)
    nz = z.con_jac_nz.cpu().numpy()
    want = dense.con_jac.cpu().numpy()
    for i in (0, N // 2, N - 1):
        A = np.zeros((mp.n + 1, mp.m))
        A.reshape(-1)[idx] = nz[i]
        np.testing.assert_array_equal(A[:mp.n], want[i])
        assert not A[mp.n].any()
    # nothing but the listed entries is ever non-zero in the dense matrix
    mask = np.ones(mp.n * mp.m, bool)
    mask[idx] = False
    assert not want.reshape(N, -1)[:, mask].any()
    only = ctx.eval_fd_nz(sc, x, want_grad=False, want_con=False)
    assert torch.equal(only.con_jac_nz, z.con_jac_nz) and torch.equal(only.cost, dense.cost)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("H", [3, 10, 20, 60])
def test_different_cost_and_constraint_time_steps(algo, H):
    """The reference's code defaults: UR20Dynamics.dt = 0.05 drives the cost rollout, URMPC.dt = 0.1 the
    constraint rollout (ur20_dynamics.py:19,97-98 / ur_mpc.py:30,705), other weights too.  The kernels keep
    two joint trajectories in that case (template SAME_DT = false)."""
    import dataclasses

    from mpc_amd.params import MpcParams

    mp = dataclasses.replace(MpcParams(), horizon=H, algo=algo, use_dynamic_model=True)
    assert mp.dt != mp.dyn_dt
    if not _supported(mp):
        pytest.skip("algo not available for these parameters")
    N = 48
    co, hs = host_scenarios(mp, N, motion="angular", cfg_id=21, with_prev=True, near_bounds=True)
    cost, grad, con, jac = co.eval_fd(hs.as_oracle_dict())
    ctx = _ctx(mp, N)
    sc, x = to_device(ctx, hs)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    _compare(res, cost, grad, con, jac)
    assert torch.equal(ctx.eval(sc, x).cost, res.cost)
    if ctx.nz_supported():
        z = ctx.eval_fd_nz(sc, x)
        assert torch.equal(ctx.expand_jacobian(z.con_jac_nz), res.con_jac) and torch.equal(z.grad, res.grad)


@pytest.mark.parametrize("H", [40, 41, 42, 43, 64, 150, 170, 171, 255])
def test_horizon_boundaries_of_the_kernel_variants(H):
    """MPC_ALGO_AUTO across the hand-overs: fused kernel (LDS image, H <= ~40), long-horizon two-kernel route
    (up to H = 170), plain rollout kernel beyond (up to MPC_MAX_HORIZON = 255)."""
    mp = params_for(H)
    N = 6
    co, hs = host_scenarios(mp, N, motion="linear", cfg_id=22, near_bounds=True)
    cost, grad, con, jac = co.eval_fd(hs.as_oracle_dict())
    ctx = _ctx(mp, N)
    sc, x = to_device(ctx, hs)
    res = ctx.eval_fd(sc, x)
    torch.cuda.synchronize()
    _compare(res, cost, grad, con, jac)
    assert ctx.nz_supported() == (H <= 170)
    with pytest.raises(ValueError):
        params_for(256)


def test_kernel_sincos_accuracy():
    """The prefix kernel's own sin/cos (kernels_prefix.hip: sincos_small) against numpy's, which is what the
    reference evaluates (ur20_dynamics.py:97-128 through the DH chain; yaw in ur_mpc.py:714-720).

    Range: the joint limits are +-2 pi and a perturbed or predicted angle stays within a few radians of them, so
    the dense part of the sample covers [-8, 8] rad including every multiple of pi/4 and pi/2 and their
    neighbours (where the quadrant changes and where one of the two results passes through zero); a second
    part goes out to +-1000 rad, far beyond any pose.  Bar: 2 ulp of the result's own magnitude inside
    [-8, 8] and an absolute 1e-15 beyond (the two-constant reduction loses relative accuracy next to zeros
    of sin/cos as |x| grows; the documented limit is ~1e5 rad), exact at 0, NaN for inf/NaN like the library."""
    mp = params_for(1)
    ctx = _ctx(mp, 1)
    rng = np.random.default_rng(5)
    grid = np.linspace(-8.0, 8.0, 400001)
    k = np.arange(-12, 13)
    special = np.concatenate([k * (np.pi / 4), np.nextafter(k * (np.pi / 4), 10.0), np.nextafter(k * (np.pi / 4), -10.0),
                              k * (np.pi / 4) + 1e-9, k * (np.pi / 4) - 1e-9, [0.0, -0.0, 1e-300, 1e-8, -1e-8]])
    near = np.concatenate([grid, special, rng.uniform(-8.0, 8.0, 200000)])
    far = rng.uniform(-1000.0, 1000.0, 200000)
    x = np.concatenate([near, far])
    sn, cs = ctx.sincos_probe(torch.as_tensor(x, device="cuda"))
    torch.cuda.synchronize()
    sn, cs = sn.cpu().numpy(), cs.cpu().numpy()
    rs, rc = np.sin(x), np.cos(x)
    n = len(near)
    for got, ref in ((sn, rs), (cs, rc)):
        err = np.abs(got[:n] - ref[:n])
        tol = 2.0 * np.spacing(np.abs(ref[:n])) + 1e-300
        # next to a zero of the function the result is ~1e-16..1e-9 and the reduction's absolute error
        # (~|k| * 2^-110) is what matters, not ulps of that tiny result
        assert np.all((err <= tol) | (err <= 4e-32 + 1e-16 * np.abs(ref[:n]))), float((err / np.spacing(np.abs(ref[:n]))).max())
        assert np.abs(got[n:] - ref[n:]).max() <= 1e-15
    assert np.all(sn[x == 0.0] == 0.0) and np.all(cs[x == 0.0] == 1.0)
    bad = torch.tensor([float("inf"), float("-inf"), float("nan")], device="cuda", dtype=torch.float64)
    sb, cb = ctx.sincos_probe(bad)
    assert torch.isnan(sb).all() and torch.isnan(cb).all()
    ctx.close()
