"""Development only: per-phase shader cycles of the QP kernels in a real batched solve (resident driver), from a
library built with tools/build_prof.sh.  usage: MPC_AMD_LIB=ab_libs/libmpc_amd_prof.so python tools/qp_phases.py [N]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, time
import numpy as np, torch

NAMES = {0: "e in place", 1: "forward substitution", 2: "QR rows", 3: "transform", 4: "NNLS (whole)", 5: "final-set rows",
         6: "column sums", 7: "back substitution + box", 8: "staging in (update)", 9: "(after BFGS)", 10: "staging / factor out",
         11: "(mark overhead)", 12: "BFGS products", 13: "ldl #1", 14: "ldl #2", 16: "nnls: dual pass", 17: "nnls: candidate",
         18: "nnls: householder apply", 19: "nnls: steps 6-11", 20: "ldl: A (both)", 21: "ldl: B (both)"}


def main():
    from mpc_amd import _ffi
    from mpc_amd.batch import MpcContext, ScenarioBatch
    from mpc_amd.device_solver import ResidentSLSQP
    from mpc_amd.params import MpcParams
    from mpc_amd.scenarios import DeviceKinematics, make_scenarios, to_device

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    mp = MpcParams.reference_yaml_defaults(horizon=20)
    ctx = MpcContext(mp, N, 0)
    lib = _ffi.load()
    hs = make_scenarios(mp, DeviceKinematics(ctx), N, motion="angular", cfg_id=3, goal_jitter=0.0)
    keep = np.nonzero(hs.active)[0]
    sc_all, _ = to_device(ctx, hs)
    sc = ctx.gather(sc_all, torch.from_numpy(keep.astype(np.int32)).to(ctx.device), ScenarioBatch.empty(ctx.device, len(keep), 20, False))
    x0 = np.tile(hs.u_init, (1, 20))[keep]
    buf = (ctypes.c_ulonglong * 32)()
    with ResidentSLSQP(ctx, capacity=len(keep), lanes=1) as dev:
        dev.solve(sc.slice(0, 8), x0[:8])
        torch.cuda.synchronize()
        lib.mpc_debug_qp_profile(buf, 1)
        t0 = time.perf_counter()
        r = dev.solve(sc, x0)
        dt = time.perf_counter() - t0
        lib.mpc_debug_qp_profile(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    top = v[[0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14]].sum()
    print(f"N={len(keep)} {dt:.3f}s  qps={r.n_qp}  cycles per QP step (team-thread 0): {top / r.n_qp:.0f}")
    for k in sorted(NAMES):
        print(f"  {NAMES[k]:28s} {v[k] / r.n_qp:10.0f} cycles/QP  {100 * v[k] / top:5.1f} %")


if __name__ == "__main__":
    main()
