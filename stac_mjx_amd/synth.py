"""Deterministic synthetic mocap generator for benchmarks and large tests.

Ground-truth motion: per clip a root random walk + yaw random walk with small roll/pitch, every
active hinge a sinusoid inside its range; marker offsets = configured offsets + N(0, (2 mm)^2);
keypoints = site positions from forward kinematics + N(0, (1 mm)^2).  Forward kinematics is
supplied by the caller (``fk(qpos[N,nq]) -> site_xpos[N,K,3]``): the engine's HIP kernel in
``bench.py``, the oracle in CPU tests.  Seeds: motion 0, noise 1, offsets 2.
"""

from __future__ import annotations

import numpy as np

from .mjcf import JNT_FREE, JNT_HINGE


def _quat_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)  # fmt: skip


def _axis_quat(axis, ang):
    q = np.zeros(ang.shape + (4,))
    q[..., 0] = np.cos(ang / 2)
    q[..., 1 + axis] = np.sin(ang / 2)
    return q


def synth_qpos(setup, n_clips: int, n_frames: int, seed: int = 0) -> np.ndarray:
    """Ground-truth qpos [C, F, nq] (float32)."""
    t = setup.tables
    rng = np.random.default_rng(seed)
    C, F = n_clips, n_frames
    q = np.broadcast_to(t.qpos0.astype(np.float64), (C, F, t.nq)).copy()
    fr = np.arange(F)[None, :]
    if t.njnt and int(t.jnt_type[0]) == JNT_FREE:
        start = np.concatenate([rng.uniform(-0.2, 0.2, (C, 2)), rng.uniform(0.03, 0.08, (C, 1))], 1)
        q[:, :, 0:3] = start[:, None, :] + np.cumsum(rng.normal(0, 0.002, (C, F, 3)), 1)
        yaw = rng.uniform(-np.pi, np.pi, (C, 1)) + np.cumsum(rng.normal(0, 0.03, (C, F)), 1)
        roll, pitch = rng.normal(0, 0.05, (C, F)), rng.normal(0, 0.05, (C, F))
        quat = _quat_mul(_quat_mul(_axis_quat(2, yaw), _axis_quat(1, pitch)), _axis_quat(0, roll))
        q[:, :, 3:7] = quat
    active = np.zeros(t.nbody, bool)
    for b in t.site_bodyid:
        while b > 0 and not active[b]:
            active[b] = True
            b = t.body_parentid[b]
    for j in range(t.njnt):
        if int(t.jnt_type[j]) != JNT_HINGE or not active[t.jnt_bodyid[j]]:
            continue
        a = int(t.jnt_qposadr[j])
        lo, hi = float(setup.lb[a]), float(setup.ub[a])
        lo, hi = max(lo, -1.0), min(hi, 1.0)
        mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
        f = rng.uniform(0.5, 2.0, (C, 1))
        ph = rng.uniform(0, 2 * np.pi, (C, 1))
        q[:, :, a] = mid + 0.35 * half * np.sin(2 * np.pi * f * fr / 50.0 + ph)
    return q.astype(np.float32)


def synth_offsets(setup, seed: int = 2) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (setup.tables.site_pos.astype(np.float64) + rng.normal(0, 0.002, setup.tables.site_pos.shape)).astype(np.float32)


def synth_keypoints(setup, fk, n_clips: int, n_frames: int, *, seed: int = 0, noise_seed: int = 1, noise=1e-3):
    """Returns (kp [C,F,3K] float32, q_true [C,F,nq] float32).  ``fk`` must use the offsets the caller chose."""
    q = synth_qpos(setup, n_clips, n_frames, seed)
    sx = np.asarray(fk(q.reshape(-1, setup.tables.nq)), dtype=np.float64).reshape(n_clips, n_frames, -1)
    rng = np.random.default_rng(noise_seed)
    kp = sx + rng.normal(0, noise, sx.shape)
    return kp.astype(np.float32), q
