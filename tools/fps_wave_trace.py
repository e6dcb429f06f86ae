"""Per-wave view of the level-1 FPS of the real clouds inside a REPLAYED cfg5shard step (diagnostic build of the
library, -DTPG_FPS_DEBUG): do the four waves of a workgroup agree on every round's pick and on the coordinates they
measure against, and do those coordinates belong to the pick?

    TPGAN_HIP_LIBRARY=.../ab_fps_debug.so python tools/fps_wave_trace.py
"""
import copy
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import _lib, configs, ops  # noqa: E402


def main():
    name, batch = "cfg5shard", 4
    reps = int(os.environ.get("REPS", "8"))
    mode = sys.argv[1] if len(sys.argv) > 1 else "replay"
    torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    has_dbg = hasattr(lib, "tpg_fps_debug_set")
    if has_dbg:
        lib.tpg_fps_debug_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
    A = configs.build_models(name, dev, seed=5, capturable=True)
    clip = configs.make_clip(name, batch=batch, seed=1, device=dev)
    T = len(clip[1])
    G, m, NW = T * batch, 256, 4
    buf = torch.zeros(G * m * NW * 4, dtype=torch.int32, device=dev)
    be = ops.backend_for(buf)
    for r in range(reps):
        M = copy.deepcopy(A[:3])
        M = (*M, tuple(torch.optim.Adam(mm.parameters(), lr=g.param_groups[0]["lr"], capturable=True)
                       for mm, g in zip((M[0], M[2], M[1]), A[3])))
        st = configs.graphed_step(name, M, clip, amp_dtype=None)
        torch.cuda.synchronize()
        buf.zero_()
        assert not has_dbg or lib.tpg_fps_debug_set(buf.data_ptr(), G, m) == 0
        torch.cuda.synchronize()
        configs.seed_host_rng(3)
        losses = st(clip[0], clip[1], 12, launch_eagerly=(mode == "body"))
        torch.cuda.synchronize()
        assert not has_dbg or lib.tpg_fps_debug_set(None, 0, 0) == 0
        d = buf.view(G, m, NW, 4).cpu()
        # the input of that launch: centres of level 0 of the real clouds, recomputed quietly
        trues = torch.cat([t.float() for t in st._keep["trues"]], 0)
        c0 = be.fps(trues.contiguous(), 1024)
        x1 = torch.gather(trues, 1, c0.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        ref = be.fps(x1, m).cpu()
        got = st._keep["plan_true_t"]["sa"][1][0].cpu()
        x1c = x1.cpu()
        bad = [int(i) for i in torch.nonzero((ref != got).any(1)).flatten()]
        print(f"run {r}: tempo_D {losses['tempo_D_loss']:.7f}; level-1 FPS of the real clouds wrong in clouds {bad}")
        for cl in bad[:3]:
            first = int(torch.nonzero(ref[cl] != got[cl]).flatten()[0])
            print(f"   cloud {cl}: first wrong pick at round {first}: ref {ref[cl, first:first + 5].tolist()} got {got[cl, first:first + 5].tolist()}")
            shown = 0
            for j in (range(1, m) if has_dbg else ()):
                olds = d[cl, j, :, 0].tolist()
                xyz = d[cl, j, :, 1:].view(torch.float32) if False else d[cl, j, :, 1:]
                same_old = len(set(olds)) == 1
                coords_ok = all(torch.equal(d[cl, j, w, 1:], x1c[cl, olds[w]].view(torch.int32)) for w in range(NW))
                if not same_old or not coords_ok:
                    print(f"      round {j}: per-wave previous pick {olds} (output says {int(got[cl, j - 1])}); "
                          f"coordinates match the pick per wave: {[bool(torch.equal(d[cl, j, w, 1:], x1c[cl, olds[w]].view(torch.int32))) for w in range(NW)]}")
                    shown += 1
                    if shown >= 4:
                        break
            if has_dbg and not shown:
                print("      every wave saw the same pick with the right coordinates in every round -> the distances / arg-max differ")
        del st


if __name__ == "__main__":
    main()
