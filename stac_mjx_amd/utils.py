"""Kinematics glue and clip batching (mirrors ``stac_mjx/utils.py:34-169,350-461``)."""

from __future__ import annotations

import numpy as np
import torch

CONTINUOUS_BATCH_OVERLAP = 10  # utils.py:18


def kinematics(mjx_model, mjx_data):
    """``utils.kinematics`` (utils.py:49-60): FK on ``mjx_data.qpos``; quaternions normalised and written back."""
    out = mjx_model.engine.fk(mjx_data.qpos.reshape(1, -1))
    return mjx_data.replace(qpos=out["qpos"][0], xpos=out["xpos"][0], xquat=out["xquat"][0], site_xpos=out["site_xpos"][0])


def com_pos(mjx_model, mjx_data):
    """``utils.com_pos`` (utils.py:63-74) contributes nothing to the fit: identity here."""
    return mjx_data


def get_site_xpos(mjx_data, site_idxs=None):
    return mjx_data.site_xpos if site_idxs is None else mjx_data.site_xpos[site_idxs]


def get_site_pos(mjx_model, site_idxs=None):
    return mjx_model.site_pos if site_idxs is None else mjx_model.site_pos[site_idxs]


def set_site_pos(mjx_model, offsets, site_idxs=None):
    """``utils.set_site_pos`` (utils.py:109-126): writes the offsets into the engine's model."""
    off = torch.as_tensor(offsets, dtype=torch.float32).reshape(-1, 3)
    mjx_model.engine.set_site_pos(off)
    return mjx_model.replace(site_pos=mjx_model.engine.get_site_pos())


def make_qs(q0, qs_to_opt, q):
    """``utils.make_qs`` (utils.py:129-144): ``(1 - mask) * q0 + mask * q``."""
    m = torch.as_tensor(qs_to_opt).to(dtype=q0.dtype, device=q0.device)
    return (1 - m) * q0 + m * q


def replace_qs(mjx_model, mjx_data, q):
    """``utils.replace_qs`` (utils.py:147-169)."""
    if q is None:
        print("optimization failed, continuing")
        return mjx_data
    return kinematics(mjx_model, mjx_data.replace(qpos=q))


def batch_kp_data(kp_data, n_frames_per_clip: int, continuous: bool = False):
    """``utils.batch_kp_data`` (utils.py:350-389): [frames, 3K] -> [clips, clip_frames, 3K]."""
    kp_data = np.asarray(kp_data) if not isinstance(kp_data, torch.Tensor) else kp_data
    n_frames = n_frames_per_clip
    total = kp_data.shape[0]
    n_batches = int(total // n_frames)
    if continuous:
        window = n_frames + CONTINUOUS_BATCH_OVERLAP
        if total < window:
            return kp_data.reshape((n_batches, window) + tuple(kp_data.shape[1:]))
        xp = torch if isinstance(kp_data, torch.Tensor) else np
        batches = [kp_data[s : s + window] for s in range(0, n_batches * n_frames, n_frames)]
        last = batches[-1]
        pad = CONTINUOUS_BATCH_OVERLAP
        # jp.pad(mode="wrap") on the last (short) window: continue with its own first rows
        reps = [last] + [last] * (pad // max(last.shape[0], 1) + 1)
        cat = xp.cat(reps, 0) if xp is torch else np.concatenate(reps, 0)
        batches[-1] = cat[: last.shape[0] + pad]
        return xp.stack(batches, 0) if xp is torch else np.stack(batches, 0)
    out = kp_data[: n_batches * n_frames]
    return out.reshape((n_batches, n_frames) + tuple(kp_data.shape[1:]))


def handle_edge_effects(ik_only_data, n_frames_per_clip: int):
    """``utils.handle_edge_effects`` (utils.py:393-461): sigmoid cross-fade of the overlapping clip ends."""
    ov = CONTINUOUS_BATCH_OVERLAP

    def crossfade(a, b, center=0.5, steepness=10.0):
        n = a.shape[0]
        x = np.linspace(0.0, 1.0, n)
        m = 0.5 * (1.0 + np.tanh(steepness * (x - center) / 2.0))
        m = m.reshape((n,) + (1,) * (a.ndim - 1))
        return (1.0 - m) * a + m * b

    def f(data):
        data = np.array(data)
        b = data.reshape((-1, n_frames_per_clip + ov) + data.shape[1:])
        for i in range(b.shape[0] - 1):
            b[i, -ov:] = crossfade(b[i, -ov:], b[i + 1, :ov])
        first = b[0]
        middle = b[1:-1, ov:]
        last = b[-1, ov:-ov]
        return np.concatenate([first, middle.reshape((-1,) + middle.shape[2:]), last], axis=0)

    for name in ("qpos", "kp_data", "xpos", "xquat", "marker_sites"):
        setattr(ik_only_data, name, f(getattr(ik_only_data, name)))
    return ik_only_data


# ---- quaternion helpers and velocity inference (stac_mjx/utils.py:172-347) -- post-processing, host numpy ----
_TOL = 1e-10


def quat_mul(quat1, quat2):
    """Hamilton product, any leading batch dims (utils.py:196-218)."""
    a, b = np.asarray(quat1), np.asarray(quat2)
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)  # fmt: skip


def quat_conj(quat):
    """[w, -i, -j, -k] (utils.py:240-256)."""
    q = np.asarray(quat)
    return np.stack([q[..., 0], -q[..., 1], -q[..., 2], -q[..., 3]], axis=-1)


def quat_diff(source, target):
    """Rotation from source to target (utils.py:259-274)."""
    return quat_mul(quat_conj(source), target)


def quat_to_axisangle(quat):
    """Axis-angle vector, angle encoded by its length (utils.py:277-299); batched over leading dims."""
    q = np.asarray(quat, dtype=np.float64)
    angle = 2 * np.arccos(np.clip(q[..., 0], -1.0, 1.0))
    small = angle < _TOL
    qn = np.where(small, 1.0, np.sin(angle / 2))
    wrapped = (angle + np.pi) % (2 * np.pi) - np.pi
    out = q[..., 1:4] / qn[..., None] * wrapped[..., None]
    return np.where(small[..., None], 0.0, out).astype(np.asarray(quat).dtype)


def compute_velocity_from_kinematics(qpos_trajectory, dt: float, freejoint: bool = True, max_qvel: float = 20.0):
    """Finite-difference qvel of one continuous clip (utils.py:302-347): last frame repeated, root gyro from the
    normalised quaternion difference, joint velocities clipped to +-max_qvel."""
    q = np.asarray(qpos_trajectory)
    q = np.concatenate([q, q[-1:]], axis=0)
    if not freejoint:
        return np.clip((q[1:] - q[:-1]) / dt, -max_qvel, max_qvel)
    qvel_joints = (q[1:, 7:] - q[:-1, 7:]) / dt
    qvel_translation = (q[1:, :3] - q[:-1, :3]) / dt
    diff = quat_diff(q[:-1, 3:7], q[1:, 3:7])
    diff = diff / np.linalg.norm(diff, axis=-1, keepdims=True)
    qvel_gyro = quat_to_axisangle(diff) / dt
    out = np.concatenate([qvel_translation, qvel_gyro, qvel_joints], axis=1)
    out[:, 6:] = np.clip(out[:, 6:], -max_qvel, max_qvel)
    return out.astype(q.dtype)
