"""Loader fuzz (CPU): a `SAMDHIP1` image that was truncated, bit-flipped or overwritten must come back from
samd_static_load as SAMD_E_IO -- or as an automaton every kernel can walk without leaving the image: all links and edge
targets inside the state table, every suffix-link chain ending at the root (transfer_state's climb, SO/sam/static_sam.py:99-101,
terminates), spill blocks inside the spill region.  Never a crash, never an allocation sized by a damaged header.

Runs against the default library; scripts/asan_cpu.sh runs the same file against the host-sanitizer build
(-fsanitize=address,undefined), where an out-of-bounds read of the loader / structural check / export aborts the process."""
import ctypes as C
import os

import numpy as np
import pytest

import samd_hip
from util import markov_stream

E_IO = -4                                  # include/samd_hip.h SAMD_E_IO
N_MUTATIONS = int(os.environ.get("SAMD_FUZZ_N", 10000))


def _codes():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "samd_hip.h")).read()
    import re
    return {k: int(v) for k, v in re.findall(r"#define\s+(SAMD_E_[A-Z]+|SAMD_OK)\s+\(?(-?\d+)\)?", hdr)}


def _load(path):
    h = C.c_void_p()
    rc = samd_hip.lib().samd_static_load(os.fsencode(path), C.byref(h))
    return rc, (samd_hip.StaticAutomaton(h) if rc == 0 else None)


def _assert_walkable(auto):
    """what the kernels rely on, re-derived from the export (independent of the loader's own check)"""
    info = auto.info()
    n = info["n_states"]
    t = auto.export()
    link, length, deg = t["link"], t["length"], t["deg"]
    assert link[0] == -1 and length[0] == 0
    assert ((link[1:] >= 0) & (link[1:] < n)).all()
    assert (length[link[1:]] < length[1:]).all()                      # every climb reaches the root
    assert (deg >= 0).all() and int(deg.sum()) == len(t["edge_tok"])
    ok = t["edge_tok"] >= 0
    assert ((t["edge_dst"][ok] >= 0) & (t["edge_dst"][ok] < n)).all()


@pytest.mark.parametrize("kind", [samd_hip.KIND_COUNT, samd_hip.KIND_ENDPOS])
def test_damaged_images_are_rejected_or_walkable(tmp_path, kind):
    codes = _codes()
    assert codes["SAMD_E_IO"] == E_IO
    rng = np.random.default_rng(17 + kind)
    # hubs (degree > 5: spill blocks), a vocabulary table, long runs, and -- ENDPOS -- the text region
    docs = [markov_stream(rng, 120, vocab=40) for _ in range(6)] + [[i] for i in range(40)]
    auto = samd_hip.StaticAutomaton.build(docs, 2, kind)
    good = str(tmp_path / "good.samd")
    auto.save(good)
    img = np.fromfile(good, dtype=np.uint8)
    rc, back = _load(good)
    assert rc == 0
    _assert_walkable(back)
    info = auto.info()
    assert info["n_spill"] > 0 and (kind == samd_hip.KIND_COUNT or info["n_text"] > 0)
    hdr_bytes = 8 + 7 * 8
    path = str(tmp_path / "fuzz.samd")
    outcomes = {"rejected": 0, "loaded": 0}
    for it in range(N_MUTATIONS):
        m = img.copy()
        how = it % 5
        if how == 0:                                                   # truncate anywhere (incl. inside the header)
            m = m[:int(rng.integers(0, len(m)))]
        elif how == 1:                                                 # one flipped bit
            i = int(rng.integers(0, len(m)))
            m[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif how == 2:                                                 # a flipped bit in the header (sizes, kind, version)
            i = int(rng.integers(0, hdr_bytes))
            m[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif how == 3:                                                 # a 32-bit word replaced by an extreme value
            i = int(rng.integers(0, len(m) // 4)) * 4
            m[i:i + 4] = np.frombuffer(np.int32(rng.choice([-1, -2, 0, 2 ** 31 - 1, -2 ** 31, 2 ** 30, info["n_states"], info["n_states"] - 1])).tobytes(), np.uint8)
        else:                                                          # a run of random bytes, or bytes appended
            if rng.random() < 0.3:
                m = np.concatenate([m, rng.integers(0, 256, int(rng.integers(1, 200)), dtype=np.uint8)])
            else:
                i = int(rng.integers(0, len(m)))
                k = int(rng.integers(1, 64))
                m[i:i + k] = rng.integers(0, 256, len(m[i:i + k]), dtype=np.uint8)
        m.tofile(path)
        rc, a = _load(path)
        if rc != 0:
            assert rc == E_IO, (it, how, rc, samd_hip.lib().samd_last_error())
            outcomes["rejected"] += 1
            continue
        outcomes["loaded"] += 1
        if it % 7 == 0 or outcomes["loaded"] < 200:
            _assert_walkable(a)                                        # (export itself walks every spill block: ASan watches it)
        if it % 50 == 0:
            a.save(str(tmp_path / "resave.samd"))
        del a
    # both outcomes must occur, or the fuzz tests nothing: truncations and header damage are rejected, a flipped count / end position loads
    assert outcomes["rejected"] > N_MUTATIONS // 5 and outcomes["loaded"] > N_MUTATIONS // 20, outcomes


def test_header_that_promises_more_than_the_file_holds_allocates_nothing(tmp_path):
    auto = samd_hip.StaticAutomaton.build([[3, 4, 5, 3, 4, 6]], 2, samd_hip.KIND_COUNT)
    good = str(tmp_path / "good.samd")
    auto.save(good)
    img = np.fromfile(good, dtype=np.uint8)
    for field in range(2, 7):                                          # n_states, n_edges, n_spill, vocab, n_text
        for val in (2 ** 62, 2 ** 40, 2 ** 31, -1):
            m = img.copy()
            m[8 + field * 8:16 + field * 8] = np.frombuffer(np.int64(val).tobytes(), np.uint8)
            p = str(tmp_path / "big.samd")
            m.tofile(p)
            rc, a = _load(p)
            if field == 3:                                             # n_edges is informational: only its sign is checked
                assert rc in (0, E_IO)
            else:
                assert rc == E_IO, (field, val)
