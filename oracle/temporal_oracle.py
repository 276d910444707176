"""CPU restatement of the temporal smoothing of simple_romp -- TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product path is romp_amd/temporal.py -> libromp_hip.so).

Follows simple_romp/romp/utils.py: LowPassFilter :203-215, OneEuroFilter :217-245, create_OneEuroFilter :257-258,
smooth_results :261-269, smooth_global_rot_matrix :188-192, batch_rodrigues / quat2mat :493-533; the matrix ->
axis-angle step is romp_oracle.rotmat_to_quat / quat_to_angle_axis (utils.py:535-682).  float32 torch on the CPU.
Pinned by tests/golden/temporal_seq.npz, produced by the reference's own functions (oracle/make_golden_temporal.py).
"""
import numpy as np
import torch

from . import romp_oracle as O


class _LowPass:
    def __init__(self):
        self.raw, self.filtered = None, None

    def step(self, value, alpha):
        s = value if self.raw is None else alpha * value + (1.0 - alpha) * self.filtered
        self.raw, self.filtered = value, s
        return s


class OneEuro:
    def __init__(self, mincutoff=1.0, beta=0.0, dcutoff=1.0, freq=30):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.x, self.dx = _LowPass(), _LowPass()

    def alpha(self, cutoff):
        te = 1.0 / self.freq
        tau = 1.0 / (2 * np.pi * cutoff)
        return 1.0 / (1.0 + tau / te)

    def step(self, x):
        prev = self.x.raw
        dx = 0.0 if prev is None else (x - prev) * self.freq
        edx = self.dx.step(dx, self.alpha(self.dcutoff))
        cutoff = self.mincutoff + self.beta * (abs(edx) if isinstance(edx, float) else torch.abs(edx))
        return self.x.step(x, self.alpha(cutoff))


def make_filters(smooth_coeff):
    return {'smpl_thetas': OneEuro(smooth_coeff, 0.7), 'cam': OneEuro(1.6, 0.7), 'smpl_betas': OneEuro(0.6, 0.7),
            'global_rot': OneEuro(smooth_coeff, 0.7)}


def rodrigues9(aa):
    """utils.py:493-533 for one axis-angle (3,) -> (9,)."""
    angle = torch.norm(aa + 1e-8, p=2)
    n = aa / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half)[None], torch.sin(half) * n])
    q = q / q.norm(p=2)
    w, x, y, z = q
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2])


def smooth(filters, thetas, betas, cam):
    """smooth_results for one person and one frame; float32 tensors (72,), (nb,), (3,)."""
    R = filters['global_rot'].step(rodrigues9(thetas[:3]))
    aa = torch.from_numpy(O.quat_to_angle_axis(O.rotmat_to_quat(R.reshape(1, 3, 3).numpy()))).reshape(3).float()
    aa[torch.isnan(aa)] = 0.0
    pose = torch.cat([aa, filters['smpl_thetas'].step(thetas[3:])])
    return pose, filters['smpl_betas'].step(betas), filters['cam'].step(cam)


def make_sequence(seed=0, frames=12, n_betas=10):
    """A jittery pose / shape / camera track (float32)."""
    g = torch.Generator().manual_seed(seed)
    base_t, base_b, base_c = 0.4 * torch.randn(72, generator=g), torch.randn(n_betas, generator=g), torch.tensor([0.8, 0.1, -0.2])
    drift = 0.05 * torch.randn(72, generator=g)
    out = []
    for f in range(frames):
        out.append((base_t + f * drift + 0.03 * torch.randn(72, generator=g), base_b + 0.05 * torch.randn(n_betas, generator=g),
                    base_c + 0.01 * torch.randn(3, generator=g)))
    return out
