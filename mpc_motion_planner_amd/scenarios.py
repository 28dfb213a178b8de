"""Seeded synthetic (start, target) pairs, sampled like MotionPlanner::sample_random_state
(mpc_solver/motionPlanner.cpp:102-114): q uniform in the margin-shrunk range, rejected while the joint-7
origin is below min_height; qd uniform in +-margin_velocity*vmax.  Eigen::Random / srand(time(0))
(examples/offline_trajectory.cpp:14) is replaced by SplitMix64 streams: problem i uses stream (seed, i)."""
import numpy as np

from . import capi

SEED = 20240001
_M64 = (1 << 64) - 1


def _splitmix64(state):
    """one SplitMix64 step on a uint64 array: returns (new_state, output)"""
    with np.errstate(over="ignore"):
        state = (state + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
        z = state.copy()
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
        z = z ^ (z >> np.uint64(31))
    return state, z


def _uniform(state, count):
    """count uniforms in [-1,1) per stream; state: uint64 [B] -> (state, [B,count])"""
    out = np.empty((state.shape[0], count))
    for c in range(count):
        state, z = _splitmix64(state)
        out[:, c] = (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0
    return state, out


def joint7_height(model, q):
    """z of the joint-7 origin (oMi[7]) for q [B,7] — numpy forward kinematics of the chain."""
    q = np.asarray(q, dtype=np.float64)
    B = q.shape[0]
    R = np.tile(np.eye(3), (B, 1, 1)); p = np.zeros((B, 3))
    R0 = np.array(model.R0).reshape(7, 3, 3); off = np.array(model.p).reshape(7, 3)
    for i in range(7):
        p = p + R @ off[i]
        c, s = np.cos(q[:, i]), np.sin(q[:, i])
        Rz = np.zeros((B, 3, 3)); Rz[:, 0, 0] = c; Rz[:, 0, 1] = -s; Rz[:, 1, 0] = s; Rz[:, 1, 1] = c; Rz[:, 2, 2] = 1
        R = R @ R0[i] @ Rz
    return p[:, 2]


def sample_states(B, margins, seed=SEED, stream_offset=0, min_height=0.05, model=None, salt=0):
    """[B,14] random feasible states. margins = (position, velocity, ...)."""
    model = model or capi.default_model()
    L = capi.default_limits()
    mp, mv = margins[0], margins[1]
    s = (1 - mp) * (L["qmax"] - L["qmin"]) / 2
    base = (int(seed) * 0x100000001B3 + int(salt)) & _M64
    with np.errstate(over="ignore"):
        state = np.uint64(base) + np.arange(stream_offset, stream_offset + B, dtype=np.uint64) * np.uint64(2)
    # velocity first, so that a stream's draws never depend on how many rejection rounds OTHER streams need
    state, u = _uniform(state, 7)
    v = mv * u * L["vmax"]                                                                # motionPlanner.cpp:113
    q = np.zeros((B, 7)); todo = np.ones(B, dtype=bool)
    for _ in range(200):
        state, u = _uniform(state, 7)
        cand = 0.5 * (u * (L["qmax"] - L["qmin"] - 2 * s) + (L["qmax"] + L["qmin"]))   # motionPlanner.cpp:107-108
        q[todo] = cand[todo]
        todo = joint7_height(model, q) < min_height                                       # motionPlanner.cpp:111
        if not todo.any():
            break
    return np.concatenate([q, v], axis=1)


def make_batch(B, margins=(0.9, 0.9, 0.5, 0.9, 0.1), seed=SEED, stream_offset=0):
    """(x0 [B,14], xf [B,14]) — margins default to examples/offline_trajectory.cpp:9."""
    x0 = sample_states(B, margins, seed, stream_offset, salt=0)
    xf = sample_states(B, margins, seed, stream_offset, salt=1)
    return x0, xf
