"""GPU: mpc_eval_fd_static (the entry point that allocates nothing, include/mpc_amd.h) through the C-ABI
against mpc_eval_fd on a MPC_ALGO_ROLLOUT context: same kernels, so every output is bit-identical; two calls
in a row, the second with a smaller input (the static scratch keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from helpers import assert_cost_close, assert_grad_close, host_scenarios, params_for
from mpc_amd import _ffi
from mpc_amd.batch import EvalResult, MpcContext
from mpc_amd.scenarios import to_device

pytestmark = pytest.mark.gpu


def _outputs(ctx, N):
    z = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float64, device=ctx.device)
    return EvalResult(z(N), z(N, ctx.n), z(N, ctx.m), z(N, ctx.n, ctx.m))


@pytest.mark.parametrize("H,N1,N2", [(20, 700, 33), (3, 257, 64), (100, 12, 5)])
def test_static_scratch_entry_point_equals_mpc_eval_fd(H, N1, N2):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    mp = params_for(H, algo=_ffi.ALGO_ROLLOUT)
    ctx = MpcContext(mp, max_scenarios=N1, device=0)
    co, hs = host_scenarios(mp, N1, motion="angular", cfg_id=31, with_prev=True, near_bounds=True)
    sc, x = to_device(ctx, hs)
    want1 = ctx.eval_fd(sc, x)
    x2 = (x[:N2] * 0.75).contiguous()
    want2 = ctx.eval_fd(sc.slice(0, N2), x2)
    torch.cuda.synchronize()
    out1, out2 = _outputs(ctx, N1), _outputs(ctx, N2)
    sc2 = sc.slice(0, N2)
    ctx.eval_fd_static(sc2, x2, out2)                        # warm: code objects loaded before the free bytes are read
    torch.cuda.synchronize()
    out2 = _outputs(ctx, N2)
    torch.cuda.synchronize()
    # hipMemGetInfo's free bytes see a hipMalloc inside the library; torch's counter only its own allocator
    before, free_before = torch.cuda.memory_allocated(ctx.device), torch.cuda.mem_get_info(ctx.device)[0]
    ctx.eval_fd_static(sc, x, out1)                          # first call
    ctx.eval_fd_static(sc2, x2, out2)                        # second call, smaller input, no sync in between
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(ctx.device) == before
    assert torch.cuda.mem_get_info(ctx.device)[0] == free_before
    for got, want in ((out1, want1), (out2, want2)):
        assert torch.equal(got.cost, want.cost) and torch.equal(got.grad, want.grad)
        assert torch.equal(got.con, want.con) and torch.equal(got.con_jac, want.con_jac)
    # and against the CPU oracle, to the north star's tolerances
    idx = np.arange(0, N1, max(1, N1 // 8))
    sub = {k: (None if v is None else v[idx]) for k, v in hs.as_oracle_dict().items()}
    cost, grad, con, jac = co.eval_fd(sub)
    assert_cost_close(out1.cost.cpu().numpy()[idx], cost)
    assert_grad_close(out1.grad.cpu().numpy()[idx], grad)
    np.testing.assert_allclose(out1.con.cpu().numpy()[idx], con, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out1.con_jac.cpu().numpy()[idx], jac, rtol=1e-9, atol=1e-10)


def test_static_scratch_serves_a_prefix_context_and_rejects_oversize():
    """The entry point ignores the context's algo (always the rollout kernels) and refuses what does not fit."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests need a real MI355X")
    H = 20
    mp = params_for(H)                       # MPC_ALGO_AUTO -> prefix kernel for mpc_eval_fd
    limit = _ffi.STATIC_SCRATCH_DOUBLES // (6 * H + 1)
    N = limit + 1
    ctx = MpcContext(mp, max_scenarios=N, device=0)
    co, hs = host_scenarios(mp, 64, motion="linear", cfg_id=32)
    sc, x = to_device(ctx, hs)
    out = _outputs(ctx, 64)
    ctx.eval_fd_static(sc, x, out)
    ref = MpcContext(params_for(H, algo=_ffi.ALGO_ROLLOUT), max_scenarios=64, device=0).eval_fd(sc, x)
    torch.cuda.synchronize()
    assert torch.equal(out.cost, ref.cost) and torch.equal(out.grad, ref.grad) and torch.equal(out.con_jac, ref.con_jac)
    pre = ctx.eval_fd(sc, x)                 # the prefix kernel agrees to the tolerances, not bit for bit
    assert_cost_close(out.cost.cpu().numpy(), pre.cost.cpu().numpy())
    assert_grad_close(out.grad.cpu().numpy(), pre.grad.cpu().numpy())
    # one scenario beyond the limit: a clear error, nothing launched
    from mpc_amd.batch import ScenarioBatch
    big = ScenarioBatch.empty(ctx.device, N, H, False)
    for t in (big.state0, big.target, big.base_traj, big.Jc):
        t.zero_()
    xb = torch.zeros(N, ctx.n, dtype=torch.float64, device=ctx.device)
    ob = EvalResult(torch.zeros(N, dtype=torch.float64, device=ctx.device), None, None, None)
    with pytest.raises(_ffi.MpcError, match="static scratch"):
        ctx.eval_fd_static(big, xb, ob)
    ok = EvalResult(torch.zeros(limit, dtype=torch.float64, device=ctx.device), None, None, None)
    ctx.eval_fd_static(big.slice(0, limit), xb[:limit], ok)     # exactly at the limit is served
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ok.cost).all())
