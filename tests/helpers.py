"""Shared test helpers: tolerances of the north star and oracle-backed scenario batches."""
import numpy as np

from mpc_amd.params import MpcParams
from mpc_amd.scenarios import make_scenarios
from oracle.oracle import COracle, OracleKinematics, ref_params_from

COST_RTOL = 1e-10      # "cost within 1e-10 relative"
GRAD_RTOL = 1e-7       # "gradient within 1e-7 relative"
GRAD_ATOL_SCALE = 1e-9  # x max|g|: FD cancellation floor (f ~ 1e2, h = 1e-3 -> ~1e-11 absolute)


def params_for(H, **kw):
    return MpcParams.reference_yaml_defaults(horizon=H, **kw)


def oracle_for(mp):
    return COracle(ref_params_from(mp))


def host_scenarios(mp, N, motion="linear", cfg_id=2, with_prev=False, near_bounds=False, **kw):
    co = oracle_for(mp)
    hs = make_scenarios(mp, OracleKinematics(co), N, motion=motion, cfg_id=cfg_id, with_prev=with_prev, **kw)
    if near_bounds:
        lim = np.tile(np.asarray(mp.v_lim), mp.horizon)
        n = mp.n
        hs.x[0::3, 0] = lim[0]                    # on the upper bound -> backward step
        hs.x[1::3, n - 1] = -lim[n - 1]           # on the lower bound -> forward step
        hs.x[2::3, n // 2] = lim[n // 2] - 5e-4   # within eps of the upper bound -> backward step
    return co, hs


def assert_cost_close(got, want):
    np.testing.assert_allclose(got, want, rtol=COST_RTOL, atol=0.0)


def assert_grad_close(got, want):
    got = np.asarray(got).reshape(want.shape)
    scale = np.max(np.abs(want), axis=-1, keepdims=True)
    err = np.abs(got - want)
    tol = GRAD_RTOL * np.abs(want) + GRAD_ATOL_SCALE * scale
    bad = err > tol
    assert not bad.any(), f"grad mismatch: max err {err.max():.3e}, worst rel {np.max(err / (np.abs(want) + 1e-300)):.3e}"
