"""Synthetic N-point x T-frame clips with the statistics of the reference's data
(no dataset ships with the reference and there is no network).

Fluid clip: N_hi points uniform in a ball whose radius gives the SPH particle spacing 0.025
(2 x particle_radius 0.0125, fluid_data_generation/sim_fluid_sequence.py:14); frames advect
with DT = 0.025 (train_step_final.py:7); the low-res cloud is a strided subset plus
N(0, 0.003^2) jitter (train_fluid/tempo_dataset.py:27,92-96).
Action clip: N(0, 0.4^2) blob with 1/8 exact duplicates (train_action/msr_dataset.py:72-74).
"""
import math

import torch

DT = 0.025
SPACING = 0.025


def fluid_ball_radius(n_hi):
    return (3.0 * n_hi * SPACING ** 3 / (4.0 * math.pi)) ** (1.0 / 3.0)


def fluid_clip(batch, n_hi=4096, ratio=8, frames=3, seed=1234, device="cpu", with_vel=False):
    """-> (lowres_pos_lst, highres_pos_lst): lists of `frames` tensors (B,N_lo,3) / (B,N_hi,3);
    with_vel: -> (low_pos, high_pos, low_vel, high_vel), the particle velocities per frame (the
    `--use_vel` inputs of tempo_gan_step)."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    R = fluid_ball_radius(n_hi)
    # uniform in a ball: direction ~ normal, radius ~ R * u^(1/3)
    d = torch.randn(batch, n_hi, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    u = torch.rand(batch, n_hi, 1, generator=g)
    pos0 = d * (R * u.pow(1.0 / 3.0))
    vel = 0.5 * torch.randn(batch, 1, 3, generator=g) + 0.05 * torch.randn(batch, n_hi, 3, generator=g)
    high, low = [], []
    for t in range(frames):
        p = pos0 + (t * DT) * vel
        high.append(p.to(device).contiguous())
        lo = p[:, ::ratio] + 0.003 * torch.randn(batch, n_hi // ratio, 3, generator=g)
        low.append(lo.to(device).contiguous())
    if with_vel:
        v = vel.expand(batch, n_hi, 3).contiguous()
        return (low, high, [v[:, ::ratio].contiguous().to(device) for _ in range(frames)],
                [v.to(device) for _ in range(frames)])
    return low, high


def action_clip(batch, n_hi=2048, ratio=16, frames=3, seed=1234, device="cpu"):
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    base = 0.4 * torch.randn(batch, n_hi, 3, generator=g)
    dup = n_hi // 8
    base[:, -dup:] = base[:, :dup]                                # exact repeats
    vel = 0.5 * torch.randn(batch, 1, 3, generator=g) + 0.1 * torch.randn(batch, n_hi, 3, generator=g)
    high, low = [], []
    for t in range(frames):
        p = base + (t * DT) * vel
        high.append(p.to(device).contiguous())
        low.append(p[:, ::ratio].contiguous().to(device))
    return low, high


def force_all_keep(sr_net):
    """Benchmark regime: final mask conv weight 0 / bias 1 => mask == 1
    => every slot survives hard masking and the generator output is (B, r*N_lo, 3)."""
    last = sr_net.filter_block.decoder[1]
    with torch.no_grad():
        last.weight.zero_()
        last.bias.fill_(1.0)
    return sr_net
