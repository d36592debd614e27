#!/usr/bin/env python3
"""Which numerics change moves which long-horizon episode: the sixteen whole episodes of tests/test_gpu_parity.py::
test_episode_reward_parity for one object, HIP simulator against the float64 oracle, for the library / warm start the ENVIRONMENT
selects -- run once per arm:
    python3 tools/attribute_bottle_outliers.py [obj]                                   shipped library
    HOIC_PLAIN_WARMSTART=1 python3 tools/attribute_bottle_outliers.py                  MuJoCo's plain warm start
    HOIC_LIB=libhoic_ieee.so python3 tools/attribute_bottle_outliers.py                IEEE division / sqrt (make ../libhoic_ieee.so)
    HOIC_LIB=libhoic_ieee.so HOIC_PLAIN_WARMSTART=1 python3 ...                        both = the round-4 numerics
Prints per-episode relative reward deviation and final |dq|, the outliers (reward 2e-3 / state 5e-3), the Newton cap hits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import hoo
from hoic_amd import lib
from hoic_amd.rl import PolicyGaussian
import episode_util as E

obj = sys.argv[1] if len(sys.argv) > 1 else "bottle"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hoo.build()
blob, cfg, ex, thresh = E.obj_setup(obj)
torch.manual_seed(3)
pol = PolicyGaussian(cfg, 32, 617).eval()
seqs, starts = E.episode_starts(N)
cache = os.path.join(ROOT, "gpurun_out", f"oracle_episodes_{obj}_{N}.npz")
if os.path.exists(cache):
    z = np.load(cache); ref = [(float(z["tot"][i]), int(z["n"][i]), z["q"][i]) for i in range(N)]
else:
    ref = [E.oracle_episode(hoo, blob, cfg, thresh, ex[seqs[i]], starts[i], pol) for i in range(N)]
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache, tot=[r[0] for r in ref], n=[r[1] for r in ref], q=np.stack([r[2] for r in ref]))
hip, diag = E.hip_episodes(blob, cfg, ex, thresh, seqs, starts, pol)
dr, dq = E.deviations(hip, ref)
print(f"{obj}: library {lib.build_id()} warm start {'plain' if os.environ.get('HOIC_PLAIN_WARMSTART') else 'shifted'}: "
      f"outliers {E.outliers(dr, dq)} worst reward {max(dr):.2e} worst state {max(dq):.2e} solver_cap_hits {diag['solver_cap_hits']} "
      f"contact_overflow {diag['contact_overflow']}")
print("  reward: " + " ".join(f"{x:.1e}" for x in dr))
print("  state:  " + " ".join(f"{x:.1e}" for x in dq))
