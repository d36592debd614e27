"""Record hipBLASLt / rocBLAS kernel selections (PyTorch TunableOp) for the GEMM shapes of the Box loop at 4096 envs,
float32 and bfloat16 update -> hoic_amd/data/tunableop_gfx950.csv.  Run on the target GPU; takes ~3 minutes.
usage: python3 tools/tune_gemms.py [out.csv]"""
import os, sys
sys.path.insert(0, '.')
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
import torch
import torch.cuda.tunable as tun
from hoic_amd import mjcf, motions, tuning
from hoic_amd.agent import AgentHandMimic
from hoic_amd.config import Config
out = sys.argv[1] if len(sys.argv) > 1 else tuning.DEFAULT_FILE
tun.enable(True); tun.tuning_enable(True)
tun.set_filename(out, insert_device_ordinal=False)
cfg = Config("box_future5_light_add_geom")
model = mjcf.load_packaged("box")
expert = motions.synthetic_expert(model, 17, 600)
for dt in ("f32", "bf16"):
    agent = AgentHandMimic(cfg, device=torch.device("cuda", 0), n_envs=4096, model="box", expert_seqs=expert, update_dtype=dt)
    for it in range(2):
        agent.optimize_policy(it, save_model=False)
    agent.eval_policy(0, max_steps=3)          # full-batch (4096-row) policy forward
    del agent
torch.cuda.synchronize()
print("selections:", len(tun.get_results()), "->", out)
