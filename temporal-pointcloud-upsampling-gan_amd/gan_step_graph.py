"""The adversarial step replayed from HIP graphs.

After the kernels were fused the eager step issues ~3400 launches (569 GEMMs at ~23 us of host
time each) and is bound by the HOST, not by the GPU (rocprof: 41 ms of kernels per 62 ms step).
`GraphedFluidStep` captures the whole iteration of `gan_step.tempo_gan_step` -- generator and
both discriminator updates, forward, backward and Adam -- into HIP graphs once and replays them.

Valid only in the STATIC regime of the step, which is checked on the device every replay:
  * `n_iter > 10` (mask loss active), gate open (`masking_loss < 0.1`),
  * the generator keeps every slot (no 999 padding => no host-side dummy replacement,
    discriminator.py:115-130, and shapes are fixed).
If a replay finds the regime violated, all state (parameters, buffers, optimizer state, RNG) is
restored from the pre-step snapshot and the step is re-run eagerly with the same random draws,
so results never depend on which path ran.  `tempo_gan_step_no_mask` has no gate and no mask
and is always static (`GraphedActionStep` is a "next" item).

Host randomness (noisy labels, label flip, rotation augmentation, point permutations) is drawn
on the host in the reference's order and fed through static device tensors; "no rotation" is a
multiplication by the identity matrix, which is exact in fp32, so one graph serves both cases.

Index work off the critical path: furthest-point sampling is a chain of npoint-1 dependent
rounds on B workgroups (0.73 ms per 4096->1024 call, six of them per step), i.e. latency on a
handful of the 256 CUs.  Every discriminator forward needs its FPS / ball-query / kNN indices,
and those depend on coordinates only, so each forward's `index_plan` is issued on a side stream
as soon as its input clouds exist (the real clouds at the very start of the step) and joined
right before use: inside the captured graph the chains become parallel branches that overlap
with the generator and discriminator GEMMs.

Concurrency inside the step: after the generator's forward, its backward and the two
discriminator updates are independent chains of short kernels; they run on three streams and
become parallel branches of the captured graph (`_phase_grads`); all optimizer steps follow
(`_phase_apply`).

Multi-GPU: with `sync.world_size > 1` the step is captured as two graphs (`_phase_grads`,
`_phase_apply`) and ONE flat gradient all-reduce over the three networks runs eagerly between
them (RCCL calls are kept out of capture).
"""
import contextlib
import gc
import os

import numpy as np
import torch

from .graph_conv import shadows, wgrad_side_stream
from .gan_step import _frozen, _autocast, _set_dummy_check, _NoSync, tempo_gan_step
from .losses import tpugan_sr_loss
from .set_abstraction import _plan_tensors, attach_plan_inverses, run_index_plan, sn_discard_prepared


def _state_tensors(modules, optims):
    ts = []
    for m in modules:
        ts += [p.data for p in m.parameters()] + [b for b in m.buffers()]
    for o in optims:
        for st in o.state.values():
            ts += [v for v in st.values() if torch.is_tensor(v)]
    return ts


def _flatten_state(modules, optims):
    """Move every state tensor (parameters, buffers, optimizer state) into ONE flat buffer per
    dtype, in place: the tensors keep their identity and become views.  The pre-step snapshot
    is then one device copy per dtype instead of one per tensor (652 of them at cfg2).
    Returns the flat buffers."""
    tensors = []      # (tensor, is_parameter)
    for m in modules:
        tensors += [(p, True) for p in m.parameters()] + [(b, False) for b in m.buffers()]
    for o in optims:
        for st in o.state.values():
            tensors += [(v, False) for v in st.values() if torch.is_tensor(v) and v.is_cuda]
    by_dtype, seen = {}, set()
    for t, is_param in tensors:
        if id(t) in seen:
            continue                      # a tensor registered twice moves once
        seen.add(id(t))
        by_dtype.setdefault(t.dtype, []).append((t, is_param))
    flats = []
    with torch.no_grad():
        for dtype, ts in by_dtype.items():
            align = 256 // ts[0][0].element_size()               # 256-byte aligned views
            offs, total = [], 0
            for t, _ in ts:
                offs.append(total)
                total += (t.numel() + align - 1) // align * align
            flat = torch.zeros(max(total, 1), dtype=dtype, device=ts[0][0].device)
            for (t, is_param), o in zip(ts, offs):
                view = flat[o:o + t.numel()].view(t.shape)
                view.copy_(t)
                if is_param:
                    t.data = view         # same Parameter object, storage = the flat buffer
                else:
                    t.set_(view)
            flats.append(flat)
    return flats


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a graph is being captured or replayed: a collection there may
    finalise an OLDER stepper's CUDAGraph objects (they sit in reference cycles), and destroying a graph /
    freeing its pool while this thread captures or launches one aborts the process or corrupts the replay
    (seen as host segfaults inside hipGraphLaunch in processes that build several steppers)."""
    was_on = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()


def _rotation_matrix_np():
    """gan_step.get_rotation_matrix (same three np.random draws, same Rz*Ry*Rx in fp32) without its
    five small torch ops: the host's share of a step is GPU idle time (tools/step_gap.py), and a
    rotating step draws 6 or 16 of these."""
    a = np.random.uniform(size=3) * 2 * np.pi
    c, s = np.cos(a), np.sin(a)
    Rx = np.array([[1., 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]], dtype=np.float32)
    Ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]], dtype=np.float32)
    Rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]], dtype=np.float32)
    return Rz @ (Ry @ Rx)


class GraphedFluidStep:
    def __init__(self, sr_net, spatial_dis, tempo_dis, optims, opt, lowres_pos_lst, highres_pos_lst,
                 furthest_distance=1.0, amp_dtype=None, sync=None, warmup=2, segmented=None):
        self.G, self.Ds, self.Dt = sr_net, spatial_dis, tempo_dis
        self.og, self.ot, self.os = optims
        for o in optims:
            if not all(g.get("capturable", True) for g in o.param_groups):   # Adam & co. expose the flag
                raise ValueError("graph capture needs optimizers built with capturable=True")
        self.opt, self.fd, self.amp = opt, furthest_distance, amp_dtype
        self.sync = sync or _NoSync()
        if segmented is None and os.environ.get("TPGAN_GRAPH_SEGMENTED"):     # measure the multi-GPU form on one GPU
            segmented = os.environ["TPGAN_GRAPH_SEGMENTED"] != "0"
        self.segmented = (self.sync.world_size > 1) if segmented is None else segmented
        if self.sync.world_size > 1 and not self.segmented:
            # the single-graph form has no place for the gradient all-reduce: ranks would silently
            # train independent models
            raise ValueError("a multi-rank step must be captured as two graphs (segmented=True): "
                             "the gradient all-reduce runs between them")
        dev = lowres_pos_lst[0].device
        self.dev, self.T, self.B = dev, len(highres_pos_lst), lowres_pos_lst[0].shape[0]
        self.low = [torch.empty_like(x) for x in lowres_pos_lst]
        self.high = [torch.empty_like(x) for x in highres_pos_lst]
        self._setup_staging(lowres_pos_lst[0].shape[1] * sr_net.upsample_ratio)
        self.report = torch.zeros(6, device=dev)
        self.viol = torch.zeros(1, device=dev)
        # index-plan streams: [0] the temporal discriminator's clouds, [1] the spatial one's (real
        # clouds at the start of the step, fake ones as soon as the generator's forward is done)
        self.sides = [torch.cuda.Stream(dev) for _ in range(2)]
        self.branch = torch.cuda.Stream(dev)       # the discriminators' updates
        # the temporal update is the branch that finishes last (tools/ab_env.py TPGAN_WHATIF_SKIP=t: 12.9 -> 10.0 ms
        # without it, 11.4 without the spatial one): TPGAN_BRANCH2_PRIO=-1 gives its stream the higher priority (A/B)
        self.branch2 = torch.cuda.Stream(dev, priority=int(os.environ.get("TPGAN_BRANCH2_PRIO", "0")))
        # round 3: two more parallel branches INSIDE the generator step -- the generator's mask head beside its
        # upsampling head (srnet.SRNet.body), and the spatial discriminator's forward (hence backward) of the generator
        # step beside the temporal one's.  TPGAN_GSTEP_BRANCHES=0: the serial form (A/B timing).
        mode = os.environ.get("TPGAN_GSTEP_BRANCHES", "0")         # "heads", "dis", "1" = both, "0" = none
        self.gstep_branches = mode in ("1", "dis")
        self.aux = torch.cuda.Stream(dev)
        self.aux2 = torch.cuda.Stream(dev)
        if mode in ("1", "heads") and hasattr(sr_net, "filter_block"):
            from .srnet import set_aux_stream
            set_aux_stream(sr_net, self.aux)
        self.use_plans = True
        self.defer_inverses = os.environ.get("TPGAN_DEFER_INVERSES", "1") != "0"
        self.wgrad_side = os.environ.get("TPGAN_WGRAD_SIDE", "0") != "0"    # measured: 13.1 -> 14.5 ms with it (a cross-stream edge per weight costs ~40 us): off
        self._keep = {}
        self._graphs = None
        self._capture(lowres_pos_lst, highres_pos_lst, warmup)

    KEYS = ["tempo_G_loss", "tempo_D_loss", "Chamfer_distance_no_norm", "masking_loss", "spatial_G_loss",
            "spatial_D_loss"]
    EAGER_UNTIL = 10          # n_iter <= 10: the mask loss is a placeholder (another regime): eager

    def _setup_staging(self, n_pred):
        """Static device tensors the host-drawn randomness of a step travels through."""
        dev, T, B = self.dev, self.T, self.B
        # host-drawn inputs of a step travel through ONE persistent pinned staging buffer per dtype
        # (an async copy from a temporary pageable tensor may read freed memory on HIP)
        nf = 4 + 9 * (2 * T + 2 * B)
        self._host_f = torch.zeros(nf, dtype=torch.float32).pin_memory()
        self._host_i = torch.zeros(T * n_pred, dtype=torch.int64).pin_memory()
        self._dev_f = torch.zeros(nf, dtype=torch.float32, device=dev)
        self._dev_i = torch.arange(n_pred, device=dev).repeat(T).contiguous()
        f = self._dev_f
        self.lab = f[:4]
        o = 4
        self.rot_fake_t = f[o:o + 9 * T].view(T, 3, 3); o += 9 * T
        self.rot_true_t = f[o:o + 9 * T].view(T, 3, 3); o += 9 * T
        self.rot_fake_s = f[o:o + 9 * B].view(B, 3, 3); o += 9 * B
        self.rot_true_s = f[o:o + 9 * B].view(B, 3, 3)
        self.perm_c = self._dev_i[:n_pred]
        self.perm_f = [self._dev_i[(i + 1) * n_pred:(i + 2) * n_pred] for i in range(T - 1)]
        eye = torch.eye(3).reshape(-1)
        self._host_f[4:] = eye.repeat(2 * T + 2 * B)
        self._host_f[:4] = torch.tensor([1.0, 0.1, 1.0, 1.0])
        self._dev_f.copy_(self._host_f)

    # ------------------------------------------------------------------ step body (capturable)
    def _fake_plans(self, update_D, fake_s_in, fake_t_in, make_fake_s, make_fake_t):
        """Index plans of every discriminator forward that looks at generated clouds, started on
        the two side streams as soon as those clouds exist: the generator step's forwards
        (`fake_s_in`, `fake_t_in`) and, if the discriminators are updated, the update's fake batch
        (`make_fake_*()`, built on the side stream) merged with the real batch's plan.  The
        searches of both forwards of a discriminator are launched together (`index_plans`): their
        FPS rounds then cost the latency of one.  -> join_fs, join_ft, plan_fs, plan_ft for the
        generator step; the update's clouds and plans go to self._keep."""
        Ds, Dt, opt, k = self.Ds, self.Dt, self.opt, self._keep
        if not self.use_plans:
            if update_D:
                k["fake_s"], k["plan_s"], k["fakes"], k["plan_t"] = make_fake_s(), None, make_fake_t(), None
            return (lambda: None), (lambda: None), None, None
        if not update_D:
            plan_fs, join_fs = run_index_plan(lambda: Ds.index_plan(fake_s_in), self.sides[1])
            plan_ft, join_ft = run_index_plan(lambda: Dt.merge_plans(Dt.index_plans([fake_t_in], opt.R)), self.sides[0])
            return join_fs, join_ft, plan_fs, plan_ft

        def both_s():
            fake_s = make_fake_s()
            mine, theirs = Ds.index_plans([fake_s_in, fake_s])
            return fake_s, attach_plan_inverses(mine), theirs

        def both_t():
            fakes = make_fake_t()
            mine, theirs = Dt.index_plans([fake_t_in, fakes], opt.R)
            return fakes, Dt.merge_plans([mine]), theirs       # (ONE pass: frames and pairs still run as segments)
        (k["fake_s"], plan_fs, upd_s), join_fs = run_index_plan(both_s, self.sides[1])
        (k["fakes"], plan_ft, upd_t), join_ft = run_index_plan(both_t, self.sides[0])
        # fake and real batch run as segments of ONE discriminator pass: one plan for both, put
        # together after the generator step's plan is out (same stream as the real batch's plan)
        # (their inverted indices -- read by the backward only -- are built BEHIND the event the update's forward waits
        # for: `_update_plan`; TPGAN_DEFER_INVERSES=0 for the A/B: 12.52 -> 12.32 ms at cfg2.  The same for the generator
        # step's own plans measured SLOWER, 12.62 -> 12.84: its backward then waits for the side streams once more)
        k["plan_s"], k["join_plan_s"] = self._update_plan(lambda: Ds.merge_plans([upd_s, k["plan_true_s"]]), self.sides[1])
        k["plan_t"], k["join_plan_t"] = self._update_plan(lambda: Dt.merge_plans([upd_t, k["plan_true_t"]]), self.sides[0])
        return join_fs, join_ft, plan_fs, plan_ft

    def _update_plan(self, merge, side):
        """The merged (fake + real) index plan of a discriminator update on `side`: the lists first, an event for the
        update's forward, then the lists' inverted indices (a row gather's backward is their only reader; the update
        waits for the whole side stream before its backward starts).  -> plan, join (to call on the update's stream)."""
        from . import ops
        if not self.defer_inverses:
            plan, _ = run_index_plan(merge, side)
            return plan, (lambda: torch.cuda.current_stream(self.dev).wait_stream(side))

        def lists_only():
            with ops.deferred_inverses() as pending:
                plan = merge()
            return plan, list(pending)
        (plan, pending), join = run_index_plan(lists_only, side)
        with torch.cuda.stream(side):
            ops.run_inverses(pending)
        return plan, join

    def _prepare_sn_in_gap(self):
        """Called (discriminators frozen) right before the generator step's stream waits for the
        fake clouds' index plans: the spectral-norm power iterations of the two forwards that follow
        run in that wait (+2.5 % steps/s).  They must stay on THIS stream: started beside the
        generator's forward on a side stream the replayed step gets 1.2-1.5 ms slower, and update
        weights made on a side stream put their autograd node there, on which hipStreamEndCapture
        segfaults (ROCm 7.2)."""
        if self.use_plans:
            sn_discard_prepared()
            self.Ds.prepare_sn(1)
            self.Dt.prepare_sn(self.T, 1)

    def _spatial_term(self, forward, join_plan, inputs, lab):
        """The generator step's spatial-discriminator term (train_step_final.py:120-122), issued FIRST (the heads'
        dropout draws keep the eager step's order) on its own stream when `gstep_branches`: it then runs -- forward
        and backward -- beside the temporal discriminator's.  -> the loss tensor (`_join_spatial_term` before use)."""
        if not self.gstep_branches:
            join_plan()
            return (0.5 * (forward().float() - lab[2]) ** 2).mean()
        main = torch.cuda.current_stream(self.dev)
        self.aux2.wait_stream(main)
        with torch.cuda.stream(self.aux2):
            join_plan()
            for t in inputs:
                t.record_stream(self.aux2)
            loss = (0.5 * (forward().float() - lab[2]) ** 2).mean()
        return loss

    def _join_spatial_term(self, loss):
        if self.gstep_branches:
            torch.cuda.current_stream(self.dev).wait_stream(self.aux2)
            loss.record_stream(torch.cuda.current_stream(self.dev))

    def _join_sides(self, stream=None):
        stream = stream or torch.cuda.current_stream(self.dev)
        for sd in self.sides + [self.aux, self.aux2]:
            stream.wait_stream(sd)

    def _seg_generator(self, update_D, defer_backward=False):
        G, Ds, Dt, opt, k = self.G, self.Ds, self.Dt, self.opt, self._keep
        low, high, lab = self.low, self.high, self.lab
        others = [0] + list(range(2, self.T))
        _set_dummy_check(Ds, False)
        _set_dummy_check(Dt, False)
        if update_D:
            # the real clouds exist already: their (rotated) copies and index plans start now
            def real_t():
                trues = [torch.matmul(h, self.rot_true_t[f]) for f, h in enumerate(high)]
                return trues, (Dt.index_plans([trues], opt.R)[0] if self.use_plans else None)

            def real_s():
                true_s = torch.bmm(high[1], self.rot_true_s)
                return true_s, (Ds.index_plans([true_s])[0] if self.use_plans else None)
            (k["trues"], k["plan_true_t"]), _ = run_index_plan(real_t, self.sides[0])
            (k["true_s"], k["plan_true_s"]), _ = run_index_plan(real_s, self.sides[1])
        # The generator has no cross-sample coupling (no BatchNorm): its T per-frame calls
        # (train_step_final.py:116-150) are ONE call on the T*B stacked clouds, centre frame first.
        order = [1] + others
        with _autocast(self.amp, self.dev):
            stacked = torch.cat([low[f] for f in order], 0)
            edge_all, mask_all = G.body(stacked, stacked)
        nT = len(order)                       # unbind: ONE stack in the backward instead of per-slice fills
        mask = mask_all[:self.B]              # (the centre frame's: the only one a loss reads)
        # the position expansion of all T frames in one call on the stacked clouds (elementwise: the same values as T calls,
        # a third of the launches in front of the fork of the fake clouds' index plans); `keep_all` = every frame kept all
        pred_all, padded_all, keep_all = G.expand_pos_static(stacked, edge_all, mask_all)       # (12.20 -> 12.08 ms, same-box A/B)
        preds = pred_all.reshape(nT, self.B, *pred_all.shape[1:]).unbind(0)
        paddeds = padded_all.reshape(nT, self.B, *padded_all.shape[1:]).unbind(0)
        pred_c, padded_c = preds[0], paddeds[0]
        fake_s_in = padded_c.index_select(1, self.perm_c).float()
        viol = ~keep_all
        with _frozen(Ds, Dt), _autocast(self.amp, self.dev):
            pred_lst = [None] * self.T
            pred_lst[1] = padded_c
            for i, f in enumerate(others):
                pred_lst[f] = paddeds[i + 1].index_select(1, self.perm_f[i])
            last_padded = paddeds[-1]
            fake_t_in = [p.float() for p in pred_lst]
        join_fs, join_ft, plan_fs, plan_ft = self._fake_plans(
            update_D, fake_s_in, fake_t_in,
            lambda: torch.bmm(last_padded.detach().float(), self.rot_fake_s),
            lambda: [torch.matmul(p.detach(), self.rot_fake_t[f]) for f, p in enumerate(fake_t_in)])
        position_loss, cd, ml = tpugan_sr_loss(100., high[1], pred_c.float(), low[1], mask.float(),
                                               opt.cutoff / self.fd, 11)
        viol = viol | ~(ml.reshape(()) < 0.1)                      # NaN counts as a violation
        with _frozen(Ds, Dt), _autocast(self.amp, self.dev):
            # this stream is about to wait ~0.5 ms for the index plans: the power iterations of the
            # two forwards below (they depend on the weights alone) fill the gap
            self._prepare_sn_in_gap()
            spatial_loss = self._spatial_term(lambda: Ds(fake_s_in, plan=plan_fs), join_fs, [fake_s_in] + _plan_tensors(plan_fs), lab)
            join_ft()
            fake = Dt.forward_passes([fake_t_in], opt.R, plan=plan_ft)[0]
            tempo_loss = (0.5 * (fake.float() - lab[3]) ** 2).mean()
            self._join_spatial_term(spatial_loss)
        sr_loss = tempo_loss + spatial_loss + opt.w * position_loss
        k.update(tempo_loss=tempo_loss.detach(), spatial_loss=spatial_loss.detach(), cd=cd.detach(), ml=ml.detach())
        self.viol.copy_(viol.float().reshape(1))
        if defer_backward:
            return sr_loss
        self.og.zero_grad(set_to_none=True)
        sr_loss.backward()
        # every side-stream branch rejoins before this segment ends (required inside a capture)
        self._join_sides()

    def _phase_grads(self, update_D):
        """Forward and backward of the whole step: three backward-heavy parts side by side.

        Once the generator's forward has produced the fake frames, three things are independent:
        the generator's own backward (through the frozen discriminators), the temporal and the
        spatial discriminator update (they read only the detached fakes and the real clouds).
        Each is a chain of many short kernels with ~5 us of dependent-launch latency between
        them, so the two discriminator updates run on their own streams: inside the captured
        graph the three chains are parallel branches that fill each other's gaps.  Order kept:
        the updates' forwards come after the generator step's discriminator forwards (spectral
        norm iterations, BatchNorm running statistics), and every optimizer step comes in
        `_phase_apply`, after the generator's backward -- which still reads the discriminators'
        parameters -- has finished."""
        with self._shadows():
            try:
                self._grads_body(update_D)
            finally:
                sn_discard_prepared()          # (nothing left unless the body raised)

    def _shadows(self):
        """Low-precision copies of every parameter, cast with ONE launch (graph_conv.shadows): made
        once the state sits in its flat buffers, refreshed at the head of every step."""
        if self.amp is None or not getattr(self, "_state", None):
            return contextlib.nullcontext()
        if getattr(self, "_shadow", None) is None:
            params = [p for m in (self.G, self.Ds, self.Dt) for p in m.parameters()]
            self._shadow = shadows(self._state, params, self.amp)
        self._shadow.refresh()
        return self._shadow

    def _grads_body(self, update_D):
        sr_loss = self._seg_generator(update_D, defer_backward=True)
        k, lab = self._keep, self.lab
        main = torch.cuda.current_stream(self.dev)
        if update_D:
            self.branch.wait_stream(main)
            self.branch2.wait_stream(main)
            if self.use_plans:                          # the updates' clouds and index lists (their inverted indices: below)
                with torch.cuda.stream(self.branch):
                    k["join_plan_s"]()
                with torch.cuda.stream(self.branch2):
                    k["join_plan_t"]()
            else:
                self.branch.wait_stream(self.sides[1])
                self.branch2.wait_stream(self.sides[0])
            # issue order = the eager step's (and the reference's, train_step_final.py:171-214): the
            # temporal update first.  The streams decide what runs where; the ISSUE order decides
            # which Philox offsets the heads' dropout draws get inside a captured graph.
            def update(dis, optim, branch, passes, plan, key, **kw):
                with torch.cuda.stream(branch):
                    for t in [x for p in passes for x in (p if isinstance(p, list) else [p])] + [lab] + _plan_tensors(plan):
                        t.record_stream(branch)
                    with _autocast(self.amp, self.dev):
                        fake, true = dis.forward_passes(passes, *kw.get("args", ()), plan=plan)
                    loss = (0.5 * ((true.float() - lab[0]) ** 2 + (fake.float() - lab[1]) ** 2)).mean()
                    optim.zero_grad(set_to_none=True)
                    branch.wait_stream(kw["side"])          # the plan's inverted indices (built behind its lists)
                    loss.backward()
                    k[key] = loss.detach()
                    k[key].record_stream(main)
            skip = os.environ.get("TPGAN_WHATIF_SKIP", "")        # timing aid (tools/ab_env.py): leave an update out
            if skip:
                k["tempo_dis_loss"] = k["spatial_dis_loss"] = torch.zeros((), device=self.dev)
            if "t" not in skip:
                update(self.Dt, self.ot, self.branch2, [k["fakes"], k["trues"]], k["plan_t"], "tempo_dis_loss", args=(self.opt.R,),
                       side=self.sides[0])
            if "s" not in skip:
                update(self.Ds, self.os, self.branch, [k["fake_s"], k["true_s"]], k["plan_s"], "spatial_dis_loss", side=self.sides[1])
        else:
            k["tempo_dis_loss"] = torch.zeros((), device=self.dev)
            k["spatial_dis_loss"] = torch.zeros((), device=self.dev)
        self.og.zero_grad(set_to_none=True)
        # the generator's weight gradients leave its backward chain: computed on an index-plan stream (idle by now) as
        # parallel leaves of the graph (graph_conv.wgrad_side_stream); the join below orders them before the optimizers
        with wgrad_side_stream(main, self.sides[0] if self.wgrad_side else None):
            sr_loss.backward()
        self._join_sides(main)                     # every branch rejoins (required inside a capture)
        if update_D:
            main.wait_stream(self.branch)
            main.wait_stream(self.branch2)

    def _phase_apply(self, update_D):
        """The three optimizer steps and the loss report."""
        self.og.step()
        if update_D:
            self.ot.step()
            self.os.step()
        k = self._keep
        self.report.copy_(torch.stack([k["tempo_loss"].reshape(()), k["tempo_dis_loss"].reshape(()),
                                       k["cd"].reshape(()), k["ml"].reshape(()),
                                       k["spatial_loss"].reshape(()), k["spatial_dis_loss"].reshape(())]).float())

    def _reduced(self, update_D):
        """Networks whose gradients exist after `_phase_grads` (for the multi-GPU all-reduce)."""
        return [self.G, self.Dt, self.Ds] if update_D else [self.G]

    def _run_eager(self, update_D):
        self._phase_grads(update_D)
        self.sync.average_grads(self._reduced(update_D))       # no-op on one GPU
        self._phase_apply(update_D)

    # ------------------------------------------------------------------ capture
    def _load(self, low, high):
        for d, s in zip(self.low, low):
            d.copy_(s)
        for d, s in zip(self.high, high):
            d.copy_(s)

    def _capture(self, low, high, warmup):
        modules, optims = (self.G, self.Ds, self.Dt), (self.og, self.ot, self.os)
        before = _state_tensors(modules, optims)
        snap = [t.clone() for t in before]
        known = {t.data_ptr() for t in before}
        rng = torch.cuda.get_rng_state(self.dev)
        self._load(low, high)
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):       # real steps on the example batch; undone below
                self._run_eager(True)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        # undo the warm-up: parameters / buffers / pre-existing optimizer state from the snapshot,
        # optimizer state created by the warm-up back to zero (= a fresh optimizer)
        torch._foreach_copy_(before, snap)
        for t in _state_tensors(modules, optims):
            if t.data_ptr() not in known:
                t.zero_()
        torch.cuda.set_rng_state(rng, self.dev)
        # all state into flat buffers BEFORE capture (addresses are baked into the graphs)
        self._state = _flatten_state(modules, optims)
        self._snap = [t.clone() for t in self._state]
        # one more eager pass so that every pointer-keyed cache (spectral-norm descriptor tables,
        # BatchNorm workspaces) is primed for the NEW addresses -- no host->device copy may happen
        # while capturing -- then back to the snapshot once more
        torch.cuda.synchronize(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self._run_eager(True)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        for d, t in zip(self._snap, self._state):
            t.copy_(d)
        torch.cuda.set_rng_state(rng, self.dev)
        torch.cuda.synchronize(self.dev)
        self._graphs = {}
        for update_D in (True, False):
            graphs, pool = [], None
            if self.segmented:
                # multi-GPU: the gradient all-reduce runs eagerly between two graphs.  The first graph ends
                # by packing every gradient it produced into ONE flat buffer, the second starts by
                # averaging and unpacking it, so all the host does in between is the collective itself.
                bucket = {}

                def grads_and_pack(u=update_D, bucket=bucket):
                    self._phase_grads(u)
                    # (the gradient tensors THIS capture writes: a later capture rebinds p.grad)
                    bucket["grads"] = [p.grad for m in self._reduced(u) for p in m.parameters() if p.grad is not None]
                    # (+ this rank's regime flag: its sum tells every rank whether ANY rank left the
                    # static regime -- the step's second collective folded into the first)
                    bucket["flat"] = torch.cat([g.reshape(-1) for g in bucket["grads"]] + [self.viol.reshape(1)])

                def unpack_and_apply(u=update_D, bucket=bucket):
                    flat, views, off = bucket["flat"], [], 0
                    flat.div_(self.sync.world_size)
                    for g in bucket["grads"]:
                        views.append(flat[off:off + g.numel()].view_as(g))
                        off += g.numel()
                    torch._foreach_copy_(bucket["grads"], views)
                    self.viol.copy_(flat[off:off + 1])
                    self._phase_apply(u)
                segs = [("grads", grads_and_pack, bucket), ("apply", unpack_and_apply, None)]
            else:
                segs = [("all", lambda u=update_D: (self._phase_grads(u), self._phase_apply(u)), None)]
            for name, fn, reduce_module in segs:
                g = torch.cuda.CUDAGraph()
                # with a process group alive, its watchdog thread polls events while we capture: only
                # THIS thread's unsafe calls may invalidate the capture then
                mode = {"capture_error_mode": "thread_local"} if self.sync.world_size > 1 else {}
                with _no_gc(), torch.cuda.graph(g, pool=pool, **mode):
                    try:
                        fn()
                    except BaseException:
                        # leave the capture joinable: an unjoined side stream turns the original
                        # error into "capturing stream has unjoined work" and poisons the stream
                        self._join_sides()                   # (index-plan streams and the generator step's branches)
                        torch.cuda.current_stream(self.dev).wait_stream(self.branch)
                        torch.cuda.current_stream(self.dev).wait_stream(self.branch2)
                        raise
                pool = g.pool()
                graphs.append((g, None if reduce_module is None else reduce_module["flat"]))
            self._graphs[update_D] = graphs
            self._keep_alive = getattr(self, "_keep_alive", []) + [dict(self._keep), segs]
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ one training step
    def _stage_host_draws(self, update_D):
        """Draw the step's host randomness in the reference's order (train_step_final.py:85-90,121,
        152,171-204) into the pinned staging buffers."""
        valid, invalid = np.random.uniform(0.8, 1.2), np.random.uniform(0.0, 0.2)
        if np.random.uniform(0.0, 1.0) < 0.03:
            valid, invalid = invalid, valid
        lab_s, lab_t = np.random.uniform(0.8, 1.2), np.random.uniform(0.8, 1.2)
        n_pred = self.perm_c.numel()
        perms = [torch.randperm(n_pred) for _ in range(self.T)]
        eye = np.eye(3, dtype=np.float32)
        rft, rtt = [eye] * self.T, [eye] * self.T
        rfs, rts = [eye] * self.B, [eye] * self.B
        if update_D:
            if np.random.uniform() > 0.7:
                rft = [_rotation_matrix_np() for _ in range(self.T)]
                rtt = [_rotation_matrix_np() for _ in range(self.T)]
            if np.random.uniform() > 0.7:
                rts = [_rotation_matrix_np() for _ in range(self.B)]
                rfs = [_rotation_matrix_np() for _ in range(self.B)]
        # (the previous step ended with a host sync, so the staging buffers are free to rewrite)
        host_f = self._host_f.numpy()                  # the pinned buffer itself
        host_f[:4] = (valid, invalid, lab_s, lab_t)
        host_f[4:] = np.stack(rft + rtt + rfs + rts).reshape(-1)
        torch.cat(perms, out=self._host_i)

    def __call__(self, lowres_pos_lst, highres_pos_lst, n_iter, freeze_D=False, launch_eagerly=False):
        """Same contract as the eager step function (without velocities); returns its loss dict.
        launch_eagerly: run the body the graphs were captured from kernel by kernel instead of
        replaying them -- same streams, same host draws, same state handling; what a replay is
        compared with in tests/test_graph_gpu.py."""
        update_D = n_iter % 2 == 0 and not freeze_D
        same_shapes = (len(lowres_pos_lst) == len(self.low) and len(highres_pos_lst) == len(self.high)
                       and all(a.shape == b.shape for a, b in zip(lowres_pos_lst, self.low))
                       and all(a.shape == b.shape for a, b in zip(highres_pos_lst, self.high)))
        if n_iter <= self.EAGER_UNTIL or not same_shapes:      # graphs are shape-specialised (a ragged last
            return self._eager(lowres_pos_lst, highres_pos_lst, n_iter, freeze_D)   # batch takes the eager step)
        np_state, cpu_rng = np.random.get_state(), torch.get_rng_state()
        cuda_rng = torch.cuda.get_rng_state(self.dev)
        self._stage_host_draws(update_D)
        self._load(lowres_pos_lst, highres_pos_lst)
        self._dev_f.copy_(self._host_f, non_blocking=True)
        self._dev_i.copy_(self._host_i, non_blocking=True)
        for d, t in zip(self._snap, self._state):                       # pre-step snapshot (18 MB, one copy
            d.copy_(t)                                                  # per dtype: the state is flat)
        marks = None
        if launch_eagerly:
            self._run_eager(update_D)
        else:
            timing = getattr(self, "timing", None)      # a list: per-step (graph-1, all-reduce, graph-2) us
            if timing is not None:
                st = torch.cuda.current_stream(self.dev)
                marks = [torch.cuda.Event(enable_timing=True)]
                marks[0].record(st)
            with _no_gc():
                for g, flat in self._graphs[update_D]:
                    g.replay()
                    if marks is not None:
                        marks.append(torch.cuda.Event(enable_timing=True))
                        marks[-1].record(st)
                    if flat is not None:
                        self.sync.sum_flat(flat)            # the ONE collective of the step's gradients
                        if marks is not None:
                            marks.append(torch.cuda.Event(enable_timing=True))
                            marks[-1].record(st)
        # multi-GPU: the decision to leave the graph path must be COLLECTIVE -- the eager step issues
        # other all-reduces than the replay -- and it is: the two-graph form has summed the flag over
        # the ranks inside its one all-reduce
        viol = self.viol
        out = torch.cat([self.report, viol.reshape(1)]).cpu().tolist()   # the step's one host sync
        if marks is not None:
            self.timing.append([1e3 * a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])])
        if out[6] != 0.0:
            # not the static regime: put everything back and take the general path with the same draws
            for d, t in zip(self._snap, self._state):
                t.copy_(d)
            np.random.set_state(np_state)
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(cuda_rng, self.dev)
            return self._eager(lowres_pos_lst, highres_pos_lst, n_iter, freeze_D)
        return {k: v for k, v in zip(self.KEYS, out[:6]) if k is not None}

    def run_body_eagerly(self, lowres_pos_lst, highres_pos_lst, update_D=True):
        """One step through the SAME body the graphs were captured from, launched kernel by kernel
        (bench.py's per-kernel timing leg; host draws stay at their last values)."""
        self._load(lowres_pos_lst, highres_pos_lst)
        self._run_eager(update_D)
        torch.cuda.synchronize(self.dev)
        return {k: v for k, v in zip(self.KEYS, self.report.cpu().tolist()) if k is not None}

    def _eager(self, low, high, n_iter, freeze_D):
        return tempo_gan_step(self.G, self.Ds, self.Dt, low, None, high, None, self.fd, self.opt, n_iter, self.og,
                              self.ot, self.os, freeze_D, sync=self.sync, amp_dtype=self.amp)


class GraphedActionStep(GraphedFluidStep):
    """`gan_step.tempo_gan_step_no_mask` (train_step_final.py:233-320) replayed from hipGraphs: the
    action-clip step has no mask head, no gate and no rotation augmentation, so it is ALWAYS in its
    static regime (the violation flag stays 0).  Same structure as the fluid step: stacked
    generator call, index plans on their own streams, fake / real batch of each discriminator
    update as segments of one pass, the three backward chains as parallel branches.

    optims = (generator, temporal-D, spatial-D) optimizers, capturable."""

    KEYS = ["tempo_G_loss", "tempo_D_loss", "Chamfer_distance_no_norm", None, "spatial_G_loss", "spatial_D_loss"]
    EAGER_UNTIL = -1

    def _setup_staging(self, n_pred):
        dev, T = self.dev, self.T
        self._host_f = torch.zeros(4, dtype=torch.float32).pin_memory()
        self._host_i = torch.zeros((T + 2) * n_pred, dtype=torch.int64).pin_memory()
        self._dev_f = torch.tensor([1.0, 0.1, 1.0, 1.0], device=dev)
        self._host_f.copy_(self._dev_f.cpu())
        self._dev_i = torch.arange(n_pred, device=dev).repeat(T + 2).contiguous()
        self.lab = self._dev_f
        v = self._dev_i.view(T + 2, n_pred)
        # randperm draws of the reference, in its order: spatial G forward, centre frame, the other
        # frames, spatial D update (train_step_final.py:249,262,270,304)
        self.perm_sg, self.perm_c, self.perm_f, self.perm_sd = v[0], v[1], [v[2 + i] for i in range(T - 1)], v[T + 1]

    def _stage_host_draws(self, update_D):
        valid, invalid = np.random.uniform(0.8, 1.2), np.random.uniform(0.0, 0.2)
        if np.random.uniform(0.0, 1.0) < 0.03:
            valid, invalid = invalid, valid
        n_pred = self.perm_c.numel()
        perm_sg = torch.randperm(n_pred)
        lab_s = np.random.uniform(0.8, 1.2)
        perms = [torch.randperm(n_pred) for _ in range(self.T)]
        lab_t = np.random.uniform(0.8, 1.2)
        perm_sd = torch.randperm(n_pred) if update_D else torch.arange(n_pred)
        self._host_f.copy_(torch.tensor([valid, invalid, lab_s, lab_t], dtype=torch.float32))
        self._host_i.copy_(torch.cat([perm_sg] + perms + [perm_sd]))

    def _seg_generator(self, update_D, defer_backward=False):
        G, Ds, Dt, opt, k = self.G, self.Ds, self.Dt, self.opt, self._keep
        low, high, lab = self.low, self.high, self.lab
        others = [0] + list(range(2, self.T))
        order = [1] + others
        if update_D:                       # the real clouds exist already: their index plans start now
            (k["trues"], k["plan_true_t"]), _ = run_index_plan(
                lambda: (list(high), Dt.index_plans([list(high)], opt.R)[0] if self.use_plans else None), self.sides[0])
            (k["true_s"], k["plan_true_s"]), _ = run_index_plan(
                lambda: (high[1], Ds.index_plans([high[1]])[0] if self.use_plans else None), self.sides[1])
        with _frozen(Ds, Dt), _autocast(self.amp, self.dev):
            edge_all = G.body(torch.cat([low[f] for f in order], 0))
            edges = edge_all.reshape(len(order), self.B, *edge_all.shape[1:]).unbind(0)
            pred_c = G.expand_pos(low[1], edges[0]).float()
            fake_s_in = pred_c.index_select(1, self.perm_sg)
            pred_lst = [None] * self.T
            pred_lst[1] = pred_c.index_select(1, self.perm_c)
            for i, f in enumerate(others):
                pred_lst[f] = G.expand_pos(low[f], edges[i + 1]).float().index_select(1, self.perm_f[i])
            join_fs, join_ft, plan_fs, plan_ft = self._fake_plans(
                update_D, fake_s_in, pred_lst, lambda: pred_c.detach().index_select(1, self.perm_sd),
                lambda: [p.detach() for p in pred_lst])
            position_loss, cd, _ = tpugan_sr_loss(0, high[1], pred_c, 0., 0., 0., 0)
            self._prepare_sn_in_gap()
            spatial_loss = self._spatial_term(lambda: Ds(fake_s_in, plan=plan_fs), join_fs, [fake_s_in] + _plan_tensors(plan_fs), lab)
            join_ft()
            fake = Dt.forward_passes([pred_lst], opt.R, plan=plan_ft)[0]
            tempo_loss = (0.5 * (fake.float() - lab[3]) ** 2).mean()
            self._join_spatial_term(spatial_loss)
        sr_loss = tempo_loss + spatial_loss + opt.w * position_loss
        k.update(tempo_loss=tempo_loss.detach(), spatial_loss=spatial_loss.detach(), cd=cd.detach(),
                 ml=torch.zeros((), device=self.dev))
        self.viol.zero_()
        if defer_backward:
            return sr_loss
        self.og.zero_grad(set_to_none=True)
        sr_loss.backward()
        self._join_sides()

    def _eager(self, low, high, n_iter, freeze_D):
        from .gan_step import tempo_gan_step_no_mask
        return tempo_gan_step_no_mask(self.G, self.Ds, self.Dt, low, high, self.opt, n_iter, self.og, self.ot,
                                      self.os, freeze_D, sync=self.sync, amp_dtype=self.amp)
