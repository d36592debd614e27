"""hipBLASLt / rocBLAS kernel selections for the GEMM shapes of the rollout and the PPO update.

PyTorch's TunableOp times every candidate kernel of a GEMM shape once and remembers the fastest; the selections for
the shapes of the release configs at 4096 envs per GPU (2048-env half-batches in the rollout, 53 248 samples in the
update, float32 and bfloat16) were recorded on an MI355X with ``tools/tune_gemms.py`` and are committed as
``hoic_amd/data/tunableop_gfx950.csv``.  Loading them costs nothing at run time (no tuning happens here); shapes that
are not in the file, or a file whose validators (PyTorch / ROCm / hipBLASLt versions, GPU architecture) do not match
the running stack, fall back to the library default.  The f32 update GEMMs go from 127 to 135 TFLOP/s with it.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_HERE, "data", "tunableop_gfx950.csv")
_loaded = None


def enable_tuned_gemms(path: str | None = None) -> bool:
    """Switch TunableOp on in read-only mode and load the committed selections.  Returns whether they were loaded.
    ``HOIC_NO_TUNED_GEMMS=1`` leaves the library defaults in place."""
    global _loaded
    import torch
    if _loaded is not None:
        return _loaded
    _loaded = False
    if os.environ.get("HOIC_NO_TUNED_GEMMS") or not torch.cuda.is_available():
        return False
    path = path or DEFAULT_FILE
    if not os.path.exists(path):
        return False
    try:
        import torch.cuda.tunable as tun
        if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":      # an explicit tuning session (tools/tune_gemms.py) keeps its own settings
            return False
        tun.enable(True)
        tun.tuning_enable(False)
        tun.record_untuned_enable(False)
        _loaded = bool(tun.read_file(path))
        if not _loaded:
            tun.enable(False)
    except Exception:                                              # an older / differently built torch: library defaults
        _loaded = False
    return _loaded
