"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported under ROS stubs by oracle/ref_shim.py) in this container.

    python -m oracle.gen_golden

The fixtures hold data only (inputs and the reference's outputs); no reference source.  The
Jacobian provider injected into the reference is this repo's own (oracle.numpy_ref.ur20_jacobian),
because MoveIt is not part of the reference tree — "parity unpinned" at that boundary.

Scenario inputs come from mpc_amd.scenarios.make_scenarios driven by the C oracle's kinematics,
i.e. the same seeded family bench.py uses.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import time

import numpy as np
from scipy.optimize._numdiff import approx_derivative

from mpc_amd.params import MpcParams
from mpc_amd.scenarios import make_scenarios
from oracle import ref_shim
from oracle.numpy_ref import base_traj_from_array
from oracle.oracle import COracle, OracleKinematics, ref_params_from

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref_for(H):
    return ref_shim.load_reference({"mpc_controller/horizon": H})


def eval_cases():
    """f, c, scipy FD grad and dc/dx of the reference at seeded points."""
    cases = []
    for H, motions, n_per, store_jac in ((1, ("linear", "angular"), 3, True), (3, ("linear", "angular"), 3, True),
                                          (20, ("linear", "angular"), 2, True), (100, ("linear",), 1, False)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        kin = OracleKinematics(COracle(ref_params_from(mp)))
        for motion in motions:
            for with_prev in (False, True):
                hs = make_scenarios(mp, kin, n_per, motion=motion, cfg_id=100 + H, with_prev=with_prev)
                for i in range(n_per):
                    x = hs.x[i].copy()
                    if i == 1:  # put some variables on / within eps of the upper bound: FD sign flip
                        lim = np.tile(np.asarray(mp.v_lim), H)
                        x[0] = lim[0]
                        x[min(7, 6 * H - 1)] = lim[min(7, 6 * H - 1)] - 5e-4
                        x[6 * H - 1] = -lim[6 * H - 1]
                    ref = _ref_for(H)
                    ref._J_cached = np.array(hs.Jc[i].reshape(6, 6))
                    if with_prev:
                        ref.previous_control = hs.u_prev[i].copy()
                    state0 = hs.state0[i].copy()
                    target = {"position": hs.target[i, :3].copy(), "orientation": hs.target[i, 3:].copy()}
                    bt = base_traj_from_array(hs.base_traj[i])
                    fun = lambda z: ref._cost_function(z, state0, target, bt)
                    con = ref._get_constraints(state0, bt)[0]["fun"]
                    bnds = ref._get_bounds(x)
                    lb = np.array([b[0] for b in bnds]); ub = np.array([b[1] for b in bnds])
                    t0 = time.time()
                    f0 = fun(x); c0 = con(x)
                    g = approx_derivative(fun, x, method="2-point", abs_step=1e-3, f0=f0, bounds=(lb, ub))
                    a = approx_derivative(con, x, method="2-point", abs_step=1e-3, f0=c0, bounds=(lb, ub)) \
                        if store_jac else None
                    print(f"H={H} {motion} prev={with_prev} i={i}: f={f0:.6f} ({time.time()-t0:.1f}s)", flush=True)
                    cases.append(dict(H=H, motion=motion, with_prev=with_prev, x=x, state0=state0,
                                      target=hs.target[i].copy(), base_traj=hs.base_traj[i].copy(),
                                      Jc=hs.Jc[i].copy(),
                                      u_prev=hs.u_prev[i].copy() if with_prev else np.zeros(6),
                                      f=f0, c=c0, grad=np.atleast_1d(g), jac=a))
    return cases


def h100_angular_cases():
    """BASELINE configs[3]'s horizon with the revolute platform: f, c, scipy's FD gradient AND dc/dx of the reference
    at H=100 (one case with previous_control and variables on / near the bounds).  ~1 minute per case."""
    H = 100
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    kin = OracleKinematics(COracle(ref_params_from(mp)))
    cases = []
    for with_prev in (False, True):
        hs = make_scenarios(mp, kin, 2, motion="angular", cfg_id=100 + H, with_prev=with_prev)
        i = 1 if with_prev else 0
        x = hs.x[i].copy()
        if with_prev:
            lim = np.tile(np.asarray(mp.v_lim), H)
            x[0] = lim[0]
            x[7] = lim[7] - 5e-4
            x[6 * H - 1] = -lim[6 * H - 1]
        ref = _ref_for(H)
        ref._J_cached = np.array(hs.Jc[i].reshape(6, 6))
        if with_prev:
            ref.previous_control = hs.u_prev[i].copy()
        state0 = hs.state0[i].copy()
        target = {"position": hs.target[i, :3].copy(), "orientation": hs.target[i, 3:].copy()}
        bt = base_traj_from_array(hs.base_traj[i])
        fun = lambda z: ref._cost_function(z, state0, target, bt)
        con = ref._get_constraints(state0, bt)[0]["fun"]
        bnds = ref._get_bounds(x)
        lb = np.array([b[0] for b in bnds]); ub = np.array([b[1] for b in bnds])
        t0 = time.time()
        f0 = fun(x); c0 = con(x)
        g = approx_derivative(fun, x, method="2-point", abs_step=1e-3, f0=f0, bounds=(lb, ub))
        a = approx_derivative(con, x, method="2-point", abs_step=1e-3, f0=c0, bounds=(lb, ub))
        print(f"H={H} angular prev={with_prev}: f={f0:.6f} ({time.time()-t0:.1f}s)", flush=True)
        cases.append(dict(H=H, motion="angular", with_prev=with_prev, x=x, state0=state0, target=hs.target[i].copy(),
                          base_traj=hs.base_traj[i].copy(), Jc=hs.Jc[i].copy(),
                          u_prev=hs.u_prev[i].copy() if with_prev else np.zeros(6), f=f0, c=c0, grad=np.atleast_1d(g), jac=a))
    cs = cases
    np.savez_compressed(
        os.path.join(OUT, "eval_H100_angular.npz"), H=np.int64(H), motion=np.array([c["motion"] for c in cs]),
        with_prev=np.array([c["with_prev"] for c in cs], dtype=np.uint8), x=np.stack([c["x"] for c in cs]),
        state0=np.stack([c["state0"] for c in cs]), target=np.stack([c["target"] for c in cs]),
        base_traj=np.stack([c["base_traj"] for c in cs]), Jc=np.stack([c["Jc"] for c in cs]),
        u_prev=np.stack([c["u_prev"] for c in cs]), f=np.array([c["f"] for c in cs]), c=np.stack([c["c"] for c in cs]),
        grad=np.stack([c["grad"] for c in cs]), jac=np.stack([c["jac"] for c in cs]))


def save_eval(cases):
    byH = {}
    for c in cases:
        byH.setdefault(c["H"], []).append(c)
    for H, cs in byH.items():
        arrs = dict(
            H=np.int64(H), motion=np.array([c["motion"] for c in cs]),
            with_prev=np.array([c["with_prev"] for c in cs], dtype=np.uint8),
            x=np.stack([c["x"] for c in cs]), state0=np.stack([c["state0"] for c in cs]),
            target=np.stack([c["target"] for c in cs]), base_traj=np.stack([c["base_traj"] for c in cs]),
            Jc=np.stack([c["Jc"] for c in cs]), u_prev=np.stack([c["u_prev"] for c in cs]),
            f=np.array([c["f"] for c in cs]), c=np.stack([c["c"] for c in cs]),
            grad=np.stack([c["grad"] for c in cs]))
        if cs[0]["jac"] is not None:
            arrs["jac"] = np.stack([c["jac"] for c in cs])  # [case][m][n] as scipy returns it
        np.savez_compressed(os.path.join(OUT, f"eval_H{H}.npz"), **arrs)


def solve_cases():
    """Full compute_control ticks of the reference (real scipy SLSQP)."""
    out = []
    for H, motion, n_s in ((1, "linear", 3), (1, "angular", 2), (3, "linear", 2), (3, "angular", 2), (20, "linear", 1)):
        mp = MpcParams.reference_yaml_defaults(horizon=H)
        kin = OracleKinematics(COracle(ref_params_from(mp)))
        hs = make_scenarios(mp, kin, n_s, motion=motion, cfg_id=200 + H, goal_jitter=0.0)
        for i in range(n_s):
            ref = _ref_for(H)
            js = {"position": hs.q0[i].copy(), "velocity": np.zeros(6)}
            ee = {"position": hs.state0[i, 12:15].copy(), "orientation": hs.state0[i, 15:18].copy()}
            tg = {"position": hs.target[i, :3].copy(), "orientation": hs.target[i, 3:].copy()}
            b = hs.base[i]
            bs = {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
                  "angular_velocity": b[9:12].copy()}
            controls = []
            sols = []
            t0 = time.time()
            for tick in range(2):  # second tick exercises warm start + previous_control
                with contextlib.redirect_stdout(io.StringIO()):
                    u = ref.compute_control(js, ee, tg, bs)
                controls.append(np.array(u))
                sols.append(np.array(ref.previous_solution) if ref.previous_solution is not None
                            else np.full(6 * H, np.nan))
            print(f"solve H={H} {motion} i={i}: u0={controls[0][:3]} ({time.time()-t0:.1f}s)", flush=True)
            out.append(dict(H=H, motion=motion, q0=hs.q0[i], ee=hs.state0[i, 12:18], target=hs.target[i],
                            base=b, control=np.stack(controls), solution=np.stack(sols)))
    return out


def save_solve(cases):
    byH = {}
    for c in cases:
        byH.setdefault(c["H"], []).append(c)
    for H, cs in byH.items():
        np.savez_compressed(
            os.path.join(OUT, f"solve_H{H}.npz"), H=np.int64(H), motion=np.array([c["motion"] for c in cs]),
            q0=np.stack([c["q0"] for c in cs]), ee=np.stack([c["ee"] for c in cs]),
            target=np.stack([c["target"] for c in cs]), base=np.stack([c["base"] for c in cs]),
            control=np.stack([c["control"] for c in cs]), solution=np.stack([c["solution"] for c in cs]))


def _solve_one_h20(args):
    """One configs[2]-shaped controller (H=20) through two compute_control ticks of the real reference."""
    motion, i, row = args
    H = 20
    ref = _ref_for(H)
    q0, ee0, target, b = row
    js = {"position": q0.copy(), "velocity": np.zeros(6)}
    ee = {"position": ee0[:3].copy(), "orientation": ee0[3:].copy()}
    tg = {"position": target[:3].copy(), "orientation": target[3:].copy()}
    bs = {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
          "angular_velocity": b[9:12].copy()}
    controls, sols = [], []
    t0 = time.time()
    for tick in range(2):
        with contextlib.redirect_stdout(io.StringIO()):
            u = ref.compute_control(js, ee, tg, bs)
        controls.append(np.array(u))
        sols.append(np.array(ref.previous_solution))
    print(f"solve H=20 {motion} row {i}: u0={controls[0][:3]} ({time.time()-t0:.1f}s)", flush=True)
    return dict(H=H, motion=motion, q0=q0, ee=ee0, target=target, base=b, control=np.stack(controls),
                solution=np.stack(sols))


def _closed_loop_h20(args):
    """`ticks` ticks of the action server's loop (mpc_action_server.py:77-135) around the real reference
    at H=20, the platform following simulate_platform_motion.py's sinusoid: joint trajectory + commands."""
    from mpc_amd.scenarios import platform_state, world_ee_pose
    from oracle.numpy_ref import ur20_fk

    motion, q0, amplitude, phase, ticks = args
    H = 20
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    ref = _ref_for(H)
    fk = lambda qq, b: world_ee_pose(np.concatenate(ur20_fk(qq))[None], b)[0]
    q = q0.copy()
    ee = fk(q, platform_state(motion, amplitude, phase, 0.0))
    target = {"position": ee[:3].copy(), "orientation": ee[3:].copy()}     # mpc_action_server.py:77
    qs, us = [q.copy()], []
    t0 = time.time()
    for t in range(ticks):
        b = platform_state(motion, amplitude, phase, t * mp.dt)[0]
        bs = {"position": b[0:3], "orientation": b[3:6], "linear_velocity": b[6:9], "angular_velocity": b[9:12]}
        with contextlib.redirect_stdout(io.StringIO()):
            u = ref.compute_control({"position": q, "velocity": np.zeros(6)},
                                    {"position": ee[:3], "orientation": ee[3:]}, target, bs)
        q = q + np.asarray(u) * mp.dt                                       # mpc_action_server.py:98,135
        ee = fk(q, platform_state(motion, amplitude, phase, (t + 1) * mp.dt))
        qs.append(q.copy()); us.append(np.array(u))
    print(f"closed loop H=20 {motion}: {ticks} ticks ({time.time()-t0:.1f}s)", flush=True)
    return dict(motion=motion, q0=q0, amplitude=amplitude, phase=phase, q=np.stack(qs), u=np.stack(us))


def h20_cases(workers=4):
    """BASELINE configs[2]'s workload pinned to the reference: the first rows of the very scenario list
    bench.py's solves leg runs (angular, cfg_id 3, goal_jitter 0) plus linear ones, two ticks each, and one
    4-tick closed-loop trajectory per motion type.  ~40 s per solve on one core -> run in a small pool."""
    import multiprocessing as mpx

    H = 20
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    kin = OracleKinematics(COracle(ref_params_from(mp)))
    jobs = []
    for motion, n_s in (("angular", 4), ("linear", 3)):
        hs = make_scenarios(mp, kin, 12, motion=motion, cfg_id=3, goal_jitter=0.0)
        rows = [i for i in range(12) if hs.active[i]][:n_s]
        jobs += [(motion, i, (hs.q0[i], hs.state0[i, 12:18], hs.target[i], hs.base[i])) for i in rows]
    loops = []
    for motion in ("angular", "linear"):
        hs = make_scenarios(mp, kin, 2, motion=motion, cfg_id=7)
        loops.append((motion, hs.q0[1], float(hs.amplitude[1]), float(hs.phase[1]), 4))
    with mpx.get_context("spawn").Pool(workers) as pool:
        loop_async = pool.map_async(_closed_loop_h20, loops)
        cs = pool.map(_solve_one_h20, jobs)
        ls = loop_async.get()
    np.savez_compressed(
        os.path.join(OUT, "solve_H20_cfg3.npz"), H=np.int64(H), motion=np.array([c["motion"] for c in cs]),
        q0=np.stack([c["q0"] for c in cs]), ee=np.stack([c["ee"] for c in cs]),
        target=np.stack([c["target"] for c in cs]), base=np.stack([c["base"] for c in cs]),
        control=np.stack([c["control"] for c in cs]), solution=np.stack([c["solution"] for c in cs]))
    np.savez_compressed(
        os.path.join(OUT, "closed_loop_H20.npz"), H=np.int64(H), motion=np.array([c["motion"] for c in ls]),
        q0=np.stack([c["q0"] for c in ls]), amplitude=np.array([c["amplitude"] for c in ls]),
        phase=np.array([c["phase"] for c in ls]), q=np.stack([c["q"] for c in ls]), u=np.stack([c["u"] for c in ls]))


def sweep_cases(workers=6):
    """BASELINE configs[4]'s workload pinned to the reference: rows of the 1,048,576-row sweep
    (mpc_amd.scenarios.make_sweep: 32 x 32 amplitude x phase lattice x 1,024 goal poses) taken from all over the
    index range, both platform families, two compute_control ticks of the real reference each."""
    import multiprocessing as mpx

    from mpc_amd.scenarios import SWEEP_ROWS, make_sweep

    H = 20
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    kin = OracleKinematics(COracle(ref_params_from(mp)))
    picks = {"linear": [0, 1023, 1024 * 37 + 5, 131072 * 3 + 77777, 1024 * 700 + 311, SWEEP_ROWS - 1],
             "angular": [1, 1024 * 33 + 512, 131072 * 5 + 4242, 1024 * 999 + 1000, SWEEP_ROWS - 1024]}
    jobs, index, motions = [], [], []
    for motion, rows in picks.items():
        for i in rows:
            hs = make_sweep(mp, kin, i, 1, motion=motion)
            if not hs.active[0]:
                continue
            jobs.append((motion, i, (hs.q0[0], hs.state0[0, 12:18], hs.target[0], hs.base[0])))
            index.append(i); motions.append(motion)
    with mpx.get_context("spawn").Pool(workers) as pool:
        cs = pool.map(_solve_one_h20, jobs)
    np.savez_compressed(
        os.path.join(OUT, "solve_H20_sweep.npz"), H=np.int64(H), motion=np.array(motions), index=np.array(index, dtype=np.int64),
        q0=np.stack([c["q0"] for c in cs]), ee=np.stack([c["ee"] for c in cs]),
        target=np.stack([c["target"] for c in cs]), base=np.stack([c["base"] for c in cs]),
        control=np.stack([c["control"] for c in cs]), solution=np.stack([c["solution"] for c in cs]))


def base_motion_cases():
    """predict_base_motion of the reference: dynamic model and 3 ticks of the stateful kinematic one."""
    rng = np.random.default_rng(77)
    H = 5
    states = []
    for t in range(3):
        states.append(dict(position=rng.uniform(-.3, .3, 3), orientation=np.array([0, 0, rng.uniform(-.4, .4)]),
                           linear_velocity=rng.uniform(-.2, .2, 3), angular_velocity=rng.uniform(-.1, .1, 3)))
    dyn = _ref_for(H)
    dyn_out = [np.array([np.concatenate([e["position"], e["orientation"], e["linear_velocity"],
                                         e["angular_velocity"]]) for e in dyn.predict_base_motion(s, H)])
               for s in states]
    kinm = ref_shim.load_reference({"mpc_controller/horizon": H,
                                    "mpc_controller/base_dynamics/use_dynamic_model": False})
    kin_out = [np.array([np.concatenate([e["position"], e["orientation"], e["linear_velocity"],
                                         e["angular_velocity"]]) for e in kinm.predict_base_motion(s, H)])
               for s in states]
    base_in = np.array([np.concatenate([s["position"], s["orientation"], s["linear_velocity"],
                                        s["angular_velocity"]]) for s in states])
    np.savez_compressed(os.path.join(OUT, "base_motion.npz"), H=np.int64(H), base=base_in,
                        dynamic=np.stack(dyn_out), kinematic=np.stack(kin_out))


def base_cmd_cases():
    """The dynamic predictor of the reference with the optional commanded velocities 'v_cmd' / 'omega_cmd' in the
    platform state (ur_mpc.py:624-625), and compute_control ticks (H=3, two each) fed such states."""
    rng = np.random.default_rng(78)
    H = 3
    mp = MpcParams.reference_yaml_defaults(horizon=H)
    kin = OracleKinematics(COracle(ref_params_from(mp)))
    hs = make_scenarios(mp, kin, 4, motion="linear", cfg_id=260, goal_jitter=0.0)
    cmds = np.array([[0.25, 0.05], [-0.1, np.nan], [np.nan, -0.08], [0.0, 0.0]])    # NaN: that key is absent
    pack = lambda tr: np.array([np.concatenate([e["position"], e["orientation"], e["linear_velocity"],
                                                 e["angular_velocity"]]) for e in tr])
    trajs, controls, sols = [], [], []
    for i in range(4):
        b = hs.base[i].copy()
        b[11] = rng.uniform(-0.05, 0.05)                      # some yaw rate as well
        hs.base[i] = b
        bs = {"position": b[0:3].copy(), "orientation": b[3:6].copy(), "linear_velocity": b[6:9].copy(),
              "angular_velocity": b[9:12].copy()}
        if not np.isnan(cmds[i, 0]):
            bs["v_cmd"] = float(cmds[i, 0])
        if not np.isnan(cmds[i, 1]):
            bs["omega_cmd"] = float(cmds[i, 1])
        ref = _ref_for(H)
        trajs.append(pack(ref.predict_base_motion(bs, H)))
        js = {"position": hs.q0[i].copy(), "velocity": np.zeros(6)}
        ee = {"position": hs.state0[i, 12:15].copy(), "orientation": hs.state0[i, 15:18].copy()}
        tg = {"position": hs.target[i, :3].copy(), "orientation": hs.target[i, 3:].copy()}
        cu, so = [], []
        for tick in range(2):
            with contextlib.redirect_stdout(io.StringIO()):
                u = ref.compute_control(js, ee, tg, bs)
            cu.append(np.array(u))
            so.append(np.array(ref.previous_solution) if ref.previous_solution is not None else np.full(6 * H, np.nan))
        controls.append(np.stack(cu))
        sols.append(np.stack(so))
    np.savez_compressed(os.path.join(OUT, "base_motion_cmd.npz"), H=np.int64(H), base=hs.base, cmd=cmds, q0=hs.q0,
                        ee=hs.state0[:, 12:18], target=hs.target, dynamic=np.stack(trajs), control=np.stack(controls),
                        solution=np.stack(sols))


def main():
    if not ref_shim.reference_available():
        sys.exit("reference tree not present: fixtures can only be generated where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    if "--h20" in sys.argv:          # only the configs[2] solve fixtures (the others stay as committed)
        h20_cases()
        return
    if "--h100" in sys.argv:         # only the configs[3] angular eval fixture
        h100_angular_cases()
        return
    if "--sweep" in sys.argv:        # only the configs[4] solve fixture
        sweep_cases()
        return
    if "--base-cmd" in sys.argv:     # only the commanded-velocity fixture
        base_cmd_cases()
        return
    base_motion_cases()
    base_cmd_cases()
    save_eval(eval_cases())
    save_solve(solve_cases())
    h20_cases()
    h100_angular_cases()
    sweep_cases()


if __name__ == "__main__":
    main()
