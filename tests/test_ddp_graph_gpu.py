"""The multi-GPU form of the graphed step (two graphs + one flat gradient all-reduce) with TWO ranks.

A one-GPU box cannot host two RCCL ranks, so both processes use cuda:0 and the collectives go over
gloo (which takes device tensors); what is exercised is everything except RCCL itself: the
two-graph split, the all-reduce of the three networks' gradients between the graphs, the
collective decision to leave the graph path, state identical across ranks after the steps."""
import os
import socket
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import tpgan_amd  # noqa: F401
    from tpgan_amd import ddp
    from tpgan_amd.gan_step_graph import GraphedFluidStep
    from tpgan_amd.set_abstraction import FluidSpatialDis, FluidTempoDis
    from tpgan_amd.srnet import SRNet
    from tpgan_amd.synthetic import fluid_clip, force_all_keep
    torch.backends.cudnn.enabled = False
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sync = ddp.GradSync()
    assert sync.world_size == world
    torch.manual_seed(100 + rank)                      # different initial weights: broadcast_state must fix that
    G = force_all_keep(SRNet(3, 128)).to(dev)
    Ds, Dt = FluidSpatialDis().to(dev), FluidTempoDis(3).to(dev)
    sync.broadcast_state(G, Ds, Dt)
    opts = tuple(torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True) for m in (G, Dt, Ds))
    opt = Namespace(use_vel=False, in_node_feats=3, cutoff=0.025, R=0.10, w=0.5)
    clips = [fluid_clip(2, 1024, 8, 3, seed=10 * rank + s, device=dev) for s in range(2)]   # per-rank data
    step = GraphedFluidStep(G, Ds, Dt, opts, opt, clips[0][0], clips[0][1], 1.0, None, sync)
    assert step.segmented and len(step._graphs[True]) == 2
    np.random.seed(7)                                   # same host draws on every rank
    torch.manual_seed(7)
    for n_iter, (low, high) in zip((12, 13, 14), (clips[0], clips[1], clips[0])):
        out = step(low, high, n_iter)
        assert all(np.isfinite(v) for v in out.values()), out
    # ONE rank leaves the static regime: its flag travels inside the gradient all-reduce, and EVERY rank
    # must restore its snapshot and take the eager step (which issues other collectives than the replay)
    eager_calls = []
    inner = step._eager
    step._eager = lambda *a, **k: (eager_calls.append(1), inner(*a, **k))[1]
    (g1, flat), second = step._graphs[True]

    class FlagAfter:                                    # this rank's flag raised as if the replay had raised it
        def replay(self):
            g1.replay()
            if rank == 1:
                flat[-1:].fill_(1.0)
    step._graphs[True] = [(FlagAfter(), flat), second]
    out = step(*clips[1], 16)
    assert len(eager_calls) == 1 and all(np.isfinite(v) for v in out.values()), (eager_calls, out)
    step._graphs[True] = [(g1, flat), second]
    out = step(*clips[0], 18)                           # and back on the graph path
    assert len(eager_calls) == 1
    # data-parallel invariant: identical parameters on every rank after averaged updates
    flat = torch.cat([p.detach().reshape(-1) for m in (G, Dt, Ds) for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), float((gathered[0] - gathered[1]).abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graph_step_keeps_ranks_identical():
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)


def _action_worker(rank, world, port):
    """The same for `GraphedActionStep` (cfg4's step: no mask, no gate, T = 8 with 28 flow embeddings at reduced size)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import tpgan_amd  # noqa: F401
    from tpgan_amd import ddp
    from tpgan_amd.gan_step_graph import GraphedActionStep
    from tpgan_amd.set_abstraction import ActionSpatialDis, ActionTempoDis
    from tpgan_amd.srnet import NoMaskSRNet
    from tpgan_amd.synthetic import action_clip
    torch.backends.cudnn.enabled = False
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sync = ddp.GradSync()
    torch.manual_seed(200 + rank)
    T = 4
    G = NoMaskSRNet(3, 128, upsample_ratio=4).to(dev)
    Ds, Dt = ActionSpatialDis().to(dev), ActionTempoDis(T).to(dev)
    sync.broadcast_state(G, Ds, Dt)
    opts = tuple(torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True) for m in (G, Dt, Ds))
    opt = Namespace(R=2.0, w=2.0)
    clips = [action_clip(2, 2048, 4, T, seed=10 * rank + s, device=dev) for s in range(2)]      # per-rank data
    step = GraphedActionStep(G, Ds, Dt, opts, opt, clips[0][0], clips[0][1], 1.0, None, sync)
    assert step.segmented and len(step._graphs[True]) == 2 and len(step._graphs[False]) == 2
    np.random.seed(9)
    torch.manual_seed(9)
    for n_iter, (low, high) in zip((12, 13, 14), (clips[0], clips[1], clips[0])):
        out = step(low, high, n_iter)
        assert all(np.isfinite(v) for v in out.values()), out
        assert (out["tempo_D_loss"] > 0) == (n_iter % 2 == 0)
    flat = torch.cat([p.detach().reshape(-1) for m in (G, Dt, Ds) for p in m.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1]), float((gathered[0] - gathered[1]).abs().max())
    # and the parameters MOVED identically from the broadcast start: the all-reduced gradients were applied
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_action_graph_step_keeps_ranks_identical():
    mp.spawn(_action_worker, args=(2, _free_port()), nprocs=2, join=True)
