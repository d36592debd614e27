"""Run only PPOLearner.update_params (GAE + 5 epochs of value and policy steps on 53248 synthetic samples): development aid
for rocprofv3 --kernel-trace --stats passes over the update's kernels.
usage: python3 tools/update_only.py [f16x3|f32|bf16] [reps] [rows] [heads=kernels|autograd] [streams=3|2|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from hoic_amd import tuning
from hoic_amd.agent import PPOLearner
from hoic_amd.config import Config
dt = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 53248
heads = sys.argv[4] if len(sys.argv) > 4 else "kernels"      # autograd: nn.Linear-shaped heads through mlp.head_linear + PyTorch's elementwise losses
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda")
if heads == "autograd":
    from hoic_amd import mlp as _M
    _M.heads_fusable = lambda *a: False
tuning.enable_tuned_gemms()
cfg = Config("box_future5_light_add_geom")
g = torch.Generator(device=dev).manual_seed(0)
T, N = rows // 4096, 4096
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
b = SimpleNamespace(states=torch.clamp(rnd(T, N, 617), -5, 5), actions=rnd(T, N, 32) * 0.1, rewards=torch.rand(T, N, device=dev),
                    masks=(torch.rand(T, N, device=dev) > 0.02).float(), next_values=torch.zeros(N, device=dev), valid=None)
torch.manual_seed(0)
L = PPOLearner(cfg, 617, 32, dev, update_dtype=dt, update_streams=streams)
L.update_params(b); torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    L.update_params(b)
torch.cuda.synchronize()
print(f"update_params {dt} heads={heads} streams={streams}: {(time.time() - t0) / reps * 1e3:.2f} ms per update ({rows} samples), losses {L.last_losses}")
