"""Where does the HOST spend its time in one bench step?  (torch.profiler, CPU side)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
b = importlib.util.module_from_spec(spec); sys.argv = ["bench.py"]; spec.loader.exec_module(b)
from torch.profiler import profile, ProfilerActivity
torch.backends.cudnn.enabled = False
dt = torch.bfloat16 if (len(sys.argv) < 2 or os.environ.get("DT", "bf16") == "bf16") else None
dev = torch.device("cuda", 0)
np.random.seed(0)
models = b.build(dev)
clips = [b.fluid_clip(8, 4096, 8, 3, seed=s, device=dev) for s in range(2)]
b.run_steps(models, clips, 2, None, dt)
torch.cuda.synchronize()
t0 = time.perf_counter(); b.run_steps(models, clips, 2, None, dt); torch.cuda.synchronize()
print("2 steps:", time.perf_counter() - t0, "s", flush=True)
with profile(activities=[ProfilerActivity.CPU]) as prof:
    b.run_steps(models, clips, 1, None, dt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60))
