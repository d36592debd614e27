import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        skip = pytest.mark.skip(reason="no GPU in this container")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_objects_between_tests():
    """Simulators / agents a test leaves behind hold HIP streams and events until the garbage collector gets to them; a
    session of ~70 GPU tests otherwise runs its later tests beside dozens of idle streams of earlier ones.  Collect after
    every test and let the device drain, so that every test starts from a quiet device."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass


@pytest.fixture(scope="session")
def box_blob():
    from hoic_amd import mjcf
    return open(mjcf.packaged_model_path("box"), "rb").read()


@pytest.fixture(scope="session")
def box_model(box_blob):
    from hoic_amd import mjcf
    return mjcf.CompiledModel.from_blob(box_blob)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import hoo
    hoo.build()
    return hoo


@pytest.fixture(scope="session")
def cfg_golden():
    return np.load(os.path.join(GOLDEN, "config_box.npz"))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def cases(z):
    out = []
    for ci in range(int(z["ncases"])):
        pre = f"c{ci}_"
        out.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    return out
