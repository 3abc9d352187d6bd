import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    """a fixture file; when tests/golden/wide_<name> exists (round 5: the reference's behaviour at drafts of 65-128 nodes, written by
    tests/golden/make_golden_wide.py with the same schema) its cases are APPENDED, so every test that follows a fixture also follows the
    wide cases.  Lists are concatenated, dicts merged key by key (lists inside concatenated)."""
    with gzip.open(os.path.join(ROOT, "tests", "golden", name), "rt") as f:
        base = json.load(f)
    wide_path = os.path.join(ROOT, "tests", "golden", "wide_" + name)
    if os.path.exists(wide_path) and os.environ.get("SAMD_TEST_WIDE_GOLDENS", "1") != "0":
        with gzip.open(wide_path, "rt") as f:
            wide = json.load(f)
        if isinstance(base, list):
            base = base + wide
        else:
            for k, v in wide.items():
                base[k] = base[k] + v if isinstance(base.get(k), list) else v
    return base


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session", autouse=True)
def poisoned_allocator():
    """SAMD_TEST_POISON=1: fill PyTorch's caching allocator with NaN bit patterns before the first GPU test, so that every later
    torch.empty() hands out NaN-filled memory instead of the zeros of a fresh hipMalloc -- a kernel (or a test) that reads padding rows,
    masked cache positions or workspaces it never wrote then fails deterministically instead of once in a while on recycled memory."""
    if os.environ.get("SAMD_TEST_POISON") != "1":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    gib = int(os.environ.get("SAMD_TEST_POISON_GIB", "48"))
    big = [torch.full((1 << 28,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(gib)]          # 1 GiB each (large pool)
    mid = [torch.full((n,), float("nan"), dtype=torch.float32, device="cuda") for n in (1 << 22, 1 << 20, 1 << 19) for _ in range(64)]
    small = [torch.full((n,), float("nan"), dtype=torch.float32, device="cuda") for n in (128, 1024, 16384, 65536) for _ in range(2048)]
    torch.cuda.synchronize()
    del big, mid, small
    yield
