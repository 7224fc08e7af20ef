import json
import os
import sys

ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT_ not in sys.path:
    sys.path.insert(0, ROOT_)
import msclip_amd                                        # noqa: E402

msclip_amd.configure_runtime()                           # the benchmark's runtime settings, before the first HIP call
import numpy as np                                       # noqa: E402
import pytest                                            # noqa: E402
import torch                                             # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def set_opt(monkeypatch, model_or_engine, **fields):
    """Replace fields of an engine's EngineOptions for the rest of the test (msclip_amd/options.py; restored at teardown)."""
    eng = model_or_engine.engine() if hasattr(model_or_engine, "engine") else model_or_engine
    monkeypatch.setattr(eng, "opt", eng.opt.replace(**fields))
    return eng


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_schema(name):
    with open(os.path.join(GOLDEN, name + ".schema.json")) as f:
        return [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(f)]


_SD_CACHE = {}


def synth_sd(name, seed=0):
    """Deterministic synthetic state_dict with the reference's key order (cached per session)."""
    from msclip_amd import synth
    key = (name, seed)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = synth.synth_state_dict(load_schema(name), seed=seed)
    return _SD_CACHE[key]


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def summarize(t):
    f = t.detach().float().flatten().cpu()
    idx = torch.linspace(0, f.numel() - 1, 64).long().clamp_(max=f.numel() - 1)   # (fp32 linspace overshoots past 2^24 elements)
    return np.concatenate([[f.mean().item(), f.abs().mean().item()], f[idx].numpy()]).astype(np.float32)


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poison_gpu_memory(request):
    """GPU tests only: before each test the caching allocator's free memory is handed back and 12 GB of it filled with NaN, so
    that whatever a test's torch.empty / a fresh engine workspace is carved from holds NaN, not the previous test's (often
    correct-looking) values.  A kernel that reads a buffer before its producer wrote it then shows up instead of passing by
    luck (round 4: the front kernel's store hazard hid behind reused workspaces).  MSCLIP_TEST_NO_POISON=1 turns it off."""
    if request.node.get_closest_marker("gpu") is not None and torch.cuda.is_available() \
            and os.environ.get("MSCLIP_TEST_NO_POISON", "0") != "1":
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            junk = torch.full((int(12e9) // 4,), float("nan"), device="cuda")
            del junk
        except RuntimeError:
            pass
    yield
