"""Where does the temporal discriminator's update differ between the eager step and the replayed body?
(VERDICT r2 item 1 / ADVICE r2: GPUTEST_r02 saw tempo_D_loss 0.50 eager vs 1.37 replay at cfg5shard, batch 2.)

For a workload and batch size, in fp32 and bf16, from the same state and host draws:
  eager   = gan_step.tempo_gan_step / _no_mask (two separate discriminator forwards per update)
  body    = the graphed stepper's body launched kernel by kernel (fake + real batch as segments of one pass)
  replay  = the graphed stepper replayed
and prints the head's logits of every discriminator forward (recorded inside `_head_fp32`), per clip, plus the
loss dictionaries.  The body and the replay must agree bit for bit (deterministic path); eager vs body shows how
far the re-organisation moves the logits at this batch size.

    python tools/diag_tempo_update.py cfg5shard 2 4        # workload, batch sizes
"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpgan_amd  # noqa: E402,F401
from tpgan_amd import configs, set_abstraction  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5shard"
    batches = [int(a) for a in sys.argv[2:] if a.isdigit()] or [2, 4]
    amps = [a for a in sys.argv[2:] if a in ("fp32", "bf16")] or ["fp32", "bf16"]
    repeat = "repeat" in sys.argv[2:]             # body and replay twice each: is the path reproducible at all?
    torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    log = []
    orig = set_abstraction._head_fp32

    def spy(fc_layers, x):
        y = orig(fc_layers, x)
        log.append((fc_layers[0].in_features, y.detach().float().reshape(-1).clone()))
        return y
    set_abstraction._head_fp32 = spy
    for batch in batches:
        for amp in [None if a == "fp32" else torch.bfloat16 for a in amps]:
            A = configs.build_models(name, dev, seed=5, capturable=True)
            for m in list(A[1].modules()) + list(A[2].modules()):
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            runs = {}
            clip = configs.make_clip(name, batch=batch, seed=1, device=dev)
            for mode in ("eager", "body", "replay") + (("body2", "replay2") if repeat else ()):
                M = copy.deepcopy(A[:3])
                M = (*M, tuple(torch.optim.Adam(m.parameters(), lr=g.param_groups[0]["lr"], capturable=True)
                               for m, g in zip((M[0], M[2], M[1]), A[3])))
                stepper = None if mode == "eager" else configs.graphed_step(name, M, clip, amp_dtype=amp)
                torch.cuda.synchronize()
                log.clear()
                configs.seed_host_rng(3)
                if mode == "eager":
                    losses = configs.eager_step(name, M, clip, 12, amp_dtype=amp)
                else:
                    losses = stepper(clip[0], clip[1], 12, launch_eagerly=mode.startswith("body"))
                torch.cuda.synchronize()
                runs[mode] = (losses, [(w, v.cpu()) for w, v in log],
                              torch.cat([p.detach().reshape(-1) for p in M[2].parameters()]).cpu(),
                              {f"{n}.{k}": v.detach().cpu().clone() for n, m in zip("G Ds Dt".split(), M[:3])
                               for k, v in m.state_dict().items()})
                del stepper
            tag = f"{name} batch {batch} {'fp32' if amp is None else 'bf16'}"
            for mode, (losses, *_rest) in runs.items():
                print(f"{tag} {mode:6s} losses {losses}")
            # the recorded forwards: eager = [Ds fake(G step), Dt fake(G step), Dt fake, Dt true, Ds fake, Ds true];
            # body = [Ds(G), Dt(G), Dt fake, Dt true, Ds fake, Ds true] as well (forward_passes heads are called per pass)
            for mode in ("eager", "body"):
                print(f"{tag} {mode:6s} logits:")
                for w, v in runs[mode][1]:
                    print(f"      head in={w}: {np.array2string(v.numpy(), precision=5)}")
            pairs = [("body", "replay")] + ([("body", "body2"), ("replay", "replay2")] if repeat else [])
            for a, b in pairs:
                sa, sb = runs[a][3], runs[b][3]
                diff = [(k, float((sa[k].float() - sb[k].float()).abs().max())) for k in sa if not torch.equal(sa[k], sb[k])]
                print(f"{tag} {a} vs {b}: {len(diff)} of {len(sa)} state tensors differ", diff[:6],
                      "| losses equal:", runs[a][0] == runs[b][0])
            d = (runs["eager"][2] - runs["body"][2]).norm() / (runs["eager"][2] - torch.cat(
                [p.detach().reshape(-1) for p in A[2].parameters()]).cpu()).norm().clamp_min(1e-30)
            print(f"{tag} eager vs body: relative L2 of the Dt update {float(d):.3e}")
    set_abstraction._head_fp32 = orig


if __name__ == "__main__":
    main()
