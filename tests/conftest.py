import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the parity tests build dozens of engines: two timed runs per tile candidate keep the suite short (the bench uses five)
os.environ.setdefault("K22_TUNE_REPS", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: parity of a kernel variant that ships switched OFF because it measured slower (the fused "
                                       "GroupNorm-apply of K22_FUSE_GN, row-major weight streams) or a heavy repeat of a gated case; they RUN by default "
                                       "(round 6: the whole suite with them takes 8 min); K22_RUN_SLOW=0 skips them for a quick pass")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("K22_RUN_SLOW", "1") != "0":
        return
    skip = pytest.mark.skip(reason="K22_RUN_SLOW=0: quick pass without the variants that ship off and the heavy repeats")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- seeded state dicts are built ONCE per session ---------------------------------------------------------------------------------
# k22.init_*_state_dict fills hundreds of tensors from one CPU generator (4.5 s for the 1/3-width UNet, 13-20 s for the 1.23 B one on a GPU
# box's host cores): round 6 found that this - re-done by nearly every test because the per-file caches held one entry and the
# parametrisations alternate - was most of the GPU suite's run time and all of its box-to-box spread (697 s on one box, 1038 s on another).
# The functions are wrapped with a byte-capped LRU keyed on their arguments; callers get a fresh OrderedDict over the SAME tensors (no test
# writes into a state dict: load_state_dict copies).
_SD_CAP_BYTES = int(os.environ.get("K22_TEST_SD_CACHE_GB", "24")) << 30


def _install_state_dict_cache():
    import collections
    import functools
    import torch
    import kandinsky2_amd as k22
    cache = collections.OrderedDict()   # key -> (value, bytes)

    def nbytes(v):
        if isinstance(v, torch.Tensor):
            return v.numel() * v.element_size()
        if isinstance(v, dict):
            return sum(nbytes(x) for x in v.values())
        if isinstance(v, (tuple, list)):
            return sum(nbytes(x) for x in v)
        return 0

    def fresh(v):
        if isinstance(v, dict):
            return type(v)(v)
        if isinstance(v, tuple):
            return tuple(fresh(x) for x in v)
        return v

    def wrap(name, fn):
        @functools.wraps(fn)
        def cached(*args, **kwargs):
            key = (name, repr(args), repr(sorted(kwargs.items())))
            if key in cache:
                cache.move_to_end(key)
                return fresh(cache[key][0])
            v = fn(*args, **kwargs)
            cache[key] = (v, nbytes(v))
            while len(cache) > 1 and sum(b for _, b in cache.values()) > _SD_CAP_BYTES:
                cache.popitem(last=False)
            return fresh(v)
        return cached

    for name in dir(k22):
        if name.startswith("init_") and name.endswith("_state_dict") and callable(getattr(k22, name)):
            setattr(k22, name, wrap(name, getattr(k22, name)))


_install_state_dict_cache()
