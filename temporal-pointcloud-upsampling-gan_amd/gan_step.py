"""One adversarial iteration (generator + both discriminators), single- or multi-GPU.

Host-side mirror of the reference's `train_step_final.py` (get_rotation_matrix :10-30,
rotate_lst :38-48, tempo_gan_step :69-230, tempo_gan_step_no_mask :233-320) with the same
argument order, the same order of host/torch RNG draws and the same loss dictionary, so a
run can be compared draw for draw with the reference.  What is organised differently:

  * the non-centre frames go through the generator body as ONE batch (`forward_frames`);
  * the discriminators are frozen (requires_grad False) while the generator's loss is
    back-propagated: the reference lets that backward fill D's .grad and then discards it
    with zero_grad (train_step_final.py:162,188,214) -- same parameters afterwards, no
    wasted weight-gradient work and no spurious gradient traffic under data parallelism;
  * the gate `ml < 0.1` (:117,:166) and the gradients are reduced over the ranks by
    `sync` (see ddp.py), so every rank takes the same branch and the same optimizer step;
  * the six `.item()` host syncs at the end (:222-229) collapse into one transfer;
  * `--use_vel`: the advection features come from ONE fused search-and-sum launch for all frames
    and samples (interpolation.interpolate_vel_lst) instead of a DGL graph per frame and sample.
"""
import contextlib

import numpy as np
import torch

from .interpolation import interpolate_vel_lst
from .losses import tpugan_sr_loss

DT = 0.025


def get_rotation_matrix(device=None):
    """Random rotation Rz*Ry*Rx with angles ~ U(0, 2pi) drawn from np.random (host RNG)."""
    a = np.random.uniform(size=3) * 2 * np.pi
    Rx = torch.tensor([[1., 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]],
                      dtype=torch.float32)
    Ry = torch.tensor([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]],
                      dtype=torch.float32)
    Rz = torch.tensor([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]],
                      dtype=torch.float32)
    return torch.matmul(Rz, torch.matmul(Ry, Rx)).to(device)


def rotate_lst(pos_lst, vel_lst=None):
    """One fresh random rotation per frame, shared by the batch and -- with `vel_lst` -- by the
    frame's velocity-like features (train_step_final.py:38-48)."""
    out, vout = [], []
    for i, pos in enumerate(pos_lst):
        r0 = get_rotation_matrix(pos.device).unsqueeze(0)
        out.append(torch.bmm(pos, r0.expand(pos.shape[0], -1, -1)))
        if vel_lst is not None:
            vout.append(torch.bmm(vel_lst[i], r0.expand(pos.shape[0], -1, -1)))
    return out if vel_lst is None else (out, vout)


def _per_sample_rotation(pos):
    R = torch.stack([get_rotation_matrix(pos.device) for _ in range(pos.shape[0])], dim=0)
    return torch.bmm(pos, R)


class _NoSync:
    """Single-process stand-in for ddp.GradSync."""
    world_size = 1

    def gate_value(self, ml):
        return ml

    def average_grads(self, module):
        pass

    def average_tensors(self, grads):
        pass

    def sum_flat(self, flat):
        pass


@contextlib.contextmanager
def _frozen(*modules):
    saved = [[p.requires_grad for p in m.parameters()] for m in modules]
    for m in modules:
        m.requires_grad_(False)
    try:
        yield
    finally:
        for m, flags in zip(modules, saved):
            for p, f in zip(m.parameters(), flags):
                p.requires_grad_(f)


def _autocast(dtype, device):
    if dtype is None or dtype == torch.float32:
        return contextlib.nullcontext()
    return torch.autocast(device_type=device.type, dtype=dtype)


def _set_dummy_check(dis, flag):
    for m in dis.modules():
        if hasattr(m, "check_dummies"):
            m.check_dummies = flag


def _report(named):
    keys = list(named)
    vals = torch.stack([named[k].detach().float().reshape(-1)[0] for k in keys]).cpu().tolist()
    return dict(zip(keys, vals))


def tempo_gan_step(sr_net, spatial_dis, tempo_dis, lowres_pos_lst, lowres_vel_lst, highres_pos_lst,
                   highres_vel_lst, furthest_distance, opt, n_iter, sr_net_optim, tempo_dis_optim,
                   spatial_dis_optim, freeze_D=False, *, sync=None, amp_dtype=None, force_gate=False):
    """Fluid step with the mask head (train_step_final.py:69-230).

    force_gate: treat the gate `ml < 0.1` as open regardless of `ml` (benchmark regime,
    SURVEY.md section 8d); everything else is unchanged."""
    use_vel = bool(getattr(opt, "use_vel", False))
    sync = sync or _NoSync()
    dev = lowres_pos_lst[1].device
    T = len(highres_pos_lst)

    def g_input(f):
        """Generator features of frame f (train_step_final.py:96-104,130-137)."""
        if use_vel and getattr(opt, "in_node_feats", 3) == 6:
            return torch.cat([lowres_pos_lst[f], lowres_vel_lst[f] * DT], dim=2)
        return lowres_pos_lst[f]

    valid = np.random.uniform(0.8, 1.2)
    invalid = np.random.uniform(0.0, 0.2)
    if np.random.uniform(0.0, 1.0) < 0.03:                          # label flip
        valid, invalid = invalid, valid

    low_c = lowres_pos_lst[1]
    with _autocast(amp_dtype, dev):
        pred_c, mask_c, padded_c = sr_net(g_input(1), low_c, hard_masking=True)
    high_c = highres_pos_lst[1]
    position_loss, cd, ml = tpugan_sr_loss(100., high_c, pred_c.float(), low_c, mask_c.float(),
                                           opt.cutoff / furthest_distance, n_iter)
    gate = force_gate or bool(sync.gate_value(ml.detach()) < 0.1)    # one host sync

    zero = torch.zeros(1, device=dev)
    if gate:
        # a cloud can only hold 999-dummies if the generator padded it
        may_pad = bool(getattr(sr_net, "last_pad_flags", [True])[0])
        _set_dummy_check(spatial_dis, may_pad)
        with _frozen(spatial_dis, tempo_dis), _autocast(amp_dtype, dev):
            fake = spatial_dis(padded_c[:, torch.randperm(padded_c.shape[1])].float())
            spatial_loss = (0.5 * (fake.float() - np.random.uniform(0.8, 1.2)) ** 2).mean()
            others = [0] + list(range(2, T))
            outs = sr_net.forward_frames([g_input(f) for f in others],
                                         [lowres_pos_lst[f] for f in others], hard_masking=True)
            pred_pos_lst = [None] * T
            pred_pos_lst[1] = padded_c
            for f, (_, _, padded) in zip(others, outs):
                pred_pos_lst[f] = padded[:, torch.randperm(padded.shape[1])]
            last_padded = outs[-1][2]                                # see the D_spatial note below
            last_may_pad = bool(sr_net.last_pad_flags[-1])
            any_pad = may_pad or any(sr_net.last_pad_flags)
            _set_dummy_check(tempo_dis, any_pad)
            if use_vel:      # advection features: real ones and their interpolation at the predictions
                gt_adv_lst, pred_adv_lst = interpolate_vel_lst(pred_pos_lst, highres_pos_lst, highres_vel_lst,
                                                               opt, furthest_distance)
                fake = tempo_dis([p.float() for p in pred_pos_lst], opt.R, feat_lst=pred_adv_lst)
            else:
                fake = tempo_dis([p.float() for p in pred_pos_lst], opt.R)
            tempo_loss = (0.5 * (fake.float() - np.random.uniform(0.8, 1.2)) ** 2).mean()
    else:
        spatial_loss, tempo_loss = zero, zero

    sr_loss = tempo_loss + spatial_loss + opt.w * position_loss
    sr_net_optim.zero_grad()
    sr_loss.backward()
    sync.average_grads(sr_net)
    sr_net_optim.step()

    if n_iter % 2 == 0 and not freeze_D and gate:
        fakes = [p.detach().float() for p in pred_pos_lst]
        trues = list(highres_pos_lst)
        fake_kw, true_kw = {}, {}
        if use_vel:
            fake_adv, true_adv = pred_adv_lst, gt_adv_lst
        if np.random.uniform() > 0.7:
            if use_vel:
                # Reference quirk kept (train_step_final.py:178): the rotated fake POSITIONS are bound
                # to a name nobody reads (`pred_pos_lst_detach`), so the temporal discriminator sees
                # un-rotated fake positions with rotated advection features; the real clouds turn
                # with their features (:179).  Same draws, same order.
                _, fake_adv = rotate_lst(fakes, fake_adv)
                trues, true_adv = rotate_lst(trues, true_adv)
            else:
                fakes = rotate_lst(fakes)
                trues = rotate_lst(trues)
        if use_vel:
            fake_kw, true_kw = {"feat_lst": fake_adv}, {"feat_lst": true_adv}
        _set_dummy_check(tempo_dis, any_pad)
        with _autocast(amp_dtype, dev):
            fake = tempo_dis(fakes, opt.R, **fake_kw)
            _set_dummy_check(tempo_dis, False)                       # real clouds never hold dummies
            true = tempo_dis(trues, opt.R, **true_kw)
        tempo_dis_loss = (0.5 * ((true.float() - valid) ** 2 + (fake.float() - invalid) ** 2)).mean()
        tempo_dis_optim.zero_grad()
        tempo_dis_loss.backward()
        sync.average_grads(tempo_dis)
        tempo_dis_optim.step()

        # The reference rebinds `padded_pred_pos_batch` inside its frame loop, so the fake
        # cloud shown to the spatial discriminator here is the LAST frame's un-permuted
        # prediction, while the real one is the centre frame (train_step_final.py:131-139,209).
        fake_cloud, true_cloud = last_padded.detach().float(), high_c
        if np.random.uniform() > 0.7:
            true_cloud = _per_sample_rotation(true_cloud)
            fake_cloud = _per_sample_rotation(fake_cloud)
        _set_dummy_check(spatial_dis, last_may_pad)
        with _autocast(amp_dtype, dev):
            fake = spatial_dis(fake_cloud)
            _set_dummy_check(spatial_dis, False)
            true = spatial_dis(true_cloud)
        spatial_dis_loss = (0.5 * ((true.float() - valid) ** 2 + (fake.float() - invalid) ** 2)).mean()
        spatial_dis_optim.zero_grad()
        spatial_dis_loss.backward()
        sync.average_grads(spatial_dis)
        spatial_dis_optim.step()
    else:
        tempo_dis_loss, spatial_dis_loss = zero, zero
    _set_dummy_check(spatial_dis, True)
    _set_dummy_check(tempo_dis, True)

    return _report({"tempo_G_loss": tempo_loss, "tempo_D_loss": tempo_dis_loss,
                    "Chamfer_distance_no_norm": cd, "masking_loss": ml,
                    "spatial_G_loss": spatial_loss, "spatial_D_loss": spatial_dis_loss})


def tempo_gan_step_no_mask(sr_net, spatial_dis, tempo_dis, lowres_pos_lst, highres_pos_lst, opt, n_iter,
                           sr_net_optim, tempo_dis_optim, spatial_dis_optim, freeze_D=False, *,
                           sync=None, amp_dtype=None):
    """Action-clip step without mask head or gate (train_step_final.py:233-320)."""
    sync = sync or _NoSync()
    dev = lowres_pos_lst[1].device
    T = len(highres_pos_lst)

    valid = np.random.uniform(0.8, 1.2)
    invalid = np.random.uniform(0.0, 0.2)
    if np.random.uniform(0.0, 1.0) < 0.03:
        valid, invalid = invalid, valid

    low_c, high_c = lowres_pos_lst[1], highres_pos_lst[1]
    with _frozen(spatial_dis, tempo_dis), _autocast(amp_dtype, dev):
        pred_c, _ = sr_net(low_c, low_c)
        pred_c = pred_c.float()
        fake = spatial_dis(pred_c[:, torch.randperm(pred_c.shape[1])])
        spatial_loss = (0.5 * (fake.float() - np.random.uniform(0.8, 1.2)) ** 2).mean()
        position_loss, cd, _ = tpugan_sr_loss(0, high_c, pred_c, 0., 0., 0., 0)
        pred_pos_lst = [None] * T
        pred_pos_lst[1] = pred_c[:, torch.randperm(pred_c.shape[1])]
        others = [0] + list(range(2, T))
        outs = sr_net.forward_frames([lowres_pos_lst[f] for f in others],
                                     [lowres_pos_lst[f] for f in others])
        for f, (p, _) in zip(others, outs):
            pred_pos_lst[f] = p.float()[:, torch.randperm(p.shape[1])]
        fake = tempo_dis(pred_pos_lst, opt.R)
        tempo_loss = (0.5 * (fake.float() - np.random.uniform(0.8, 1.2)) ** 2).mean()

    sr_loss = tempo_loss + spatial_loss + opt.w * position_loss
    sr_net_optim.zero_grad()
    sr_loss.backward()
    sync.average_grads(sr_net)
    sr_net_optim.step()

    zero = torch.zeros(1, device=dev)
    if n_iter % 2 == 0 and not freeze_D:
        fakes = [p.detach() for p in pred_pos_lst]
        with _autocast(amp_dtype, dev):
            fake = tempo_dis(fakes, opt.R)
            true = tempo_dis(list(highres_pos_lst), opt.R)
        tempo_dis_loss = (0.5 * ((true.float() - valid) ** 2 + (fake.float() - invalid) ** 2)).mean()
        tempo_dis_optim.zero_grad()
        tempo_dis_loss.backward()
        sync.average_grads(tempo_dis)
        tempo_dis_optim.step()

        with _autocast(amp_dtype, dev):
            fake = spatial_dis(pred_c[:, torch.randperm(pred_c.shape[1])].detach())
            true = spatial_dis(high_c)
        spatial_dis_loss = (0.5 * ((true.float() - valid) ** 2 + (fake.float() - invalid) ** 2)).mean()
        spatial_dis_optim.zero_grad()
        spatial_dis_loss.backward()
        sync.average_grads(spatial_dis)
        spatial_dis_optim.step()
    else:
        tempo_dis_loss, spatial_dis_loss = zero, zero

    return _report({"tempo_G_loss": tempo_loss, "tempo_D_loss": tempo_dis_loss,
                    "Chamfer_distance_no_norm": cd, "spatial_G_loss": spatial_loss,
                    "spatial_D_loss": spatial_dis_loss})
