"""The native streaming reader of reference pickles (include/samd_hip.h samd_static_from_pickle; csrc/sam_pickle.cpp), CPU only.

A pickle written by the reference's own dump_sam (tests/golden/ref_static_sam.pkl) and pickles written here with stand-in classes under
the reference's module paths (every protocol 2-5, both variants, hubs with spill blocks) must give the state table pickle.load +
StaticSAM.__setstate__ gives, and the oracle's automaton of the same documents; nothing in a pickle is ever executed; anything outside
the supported opcode subset is declined with SAMD_E_IO (load_sam then falls back to pickle.load).  With /root/reference present (dev
container) a 2^20-token automaton built, dumped and pickled BY THE REFERENCE is imported, compared with the oracle and its wall time /
peak memory recorded."""
import ctypes as C
import json
import os
import pickle
import subprocess
import sys
import time
from dataclasses import dataclass
from typing import Dict

import numpy as np
import pytest

import samd_hip
from oracle import sam_oracle as O
from util import markov_stream

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def tables(auto):
    e = auto.export()
    return {k: np.asarray(v) for k, v in e.items()}


def same_tables(a, b):
    for k in ("link", "length", "aux", "deg", "edge_tok", "edge_dst"):
        assert np.array_equal(a[k], b[k]), k


def test_the_references_own_pickle(golden):
    path = os.path.join(HERE, "golden", "ref_static_sam.pkl")
    meta = golden("ref_static_sam_docs.json.gz")
    auto, pickled = samd_hip.StaticAutomaton.from_reference_pickle(path, samd_hip.KIND_COUNT)
    assert auto.info()["n_states"] == meta["n_states"]
    assert pickled["max_predicts"] is not None and pickled["alpha"] is not None and pickled["K"] == 8
    # (a) the object-graph route: pickle.load resolves the reference's class paths to ours, __setstate__ converts
    import samd_sam_only  # noqa: F401
    with open(path, "rb") as f:
        sam = pickle.load(f)
    same_tables(tables(auto), tables(sam._auto))
    # (b) the builder on the same documents (== the oracle, tests/test_builder_cpu.py)
    built = samd_hip.StaticAutomaton.build(meta["docs"], meta["eos"], samd_hip.KIND_COUNT)
    same_tables(tables(auto), tables(built))
    # (c) load_sam takes the native route without a warning
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s2 = samd_sam_only.load_sam(path)
    same_tables(tables(s2._auto), tables(built))
    assert s2.max_predicts == int(pickled["max_predicts"]) and s2.alpha == pickled["alpha"]


def reference_like_object(kind, docs, eos, monkeypatch):
    """the object graph the reference's dump_sam pickles, rebuilt from the oracle's automaton, under the reference's class paths"""
    ora = (O.StaticSAM if kind == 0 else O.StaticSAMFull).build(docs, eos)
    e = ora.export()
    mod_name = "samd_sam_only.sam.static_sam" if kind == 0 else "samd.sam.static_sam"
    import importlib
    mod = importlib.import_module(mod_name)
    aux_name = "cnt_endpos" if kind == 0 else "min_endpos"

    class StaticSAM:
        pass
    fields = {"next": Dict[int, int], "link": int, "length": int, aux_name: int}
    SAMState = dataclass(type("SAMState", (), {"__annotations__": fields}))
    SAMState.__module__ = StaticSAM.__module__ = mod_name
    SAMState.__qualname__ = "StaticSAM.SAMState"
    StaticSAM.__qualname__ = "StaticSAM"
    StaticSAM.SAMState = SAMState
    monkeypatch.setattr(mod, "StaticSAM", StaticSAM)               # pickle checks that the name resolves to the class being dumped
    sam = StaticSAM()
    states, k = [], 0
    for i in range(len(e["link"])):
        d = int(e["deg"][i])
        nxt = dict(zip(e["edge_tok"][k:k + d].tolist(), e["edge_dst"][k:k + d].tolist()))
        k += d
        states.append(SAMState(nxt, int(e["link"][i]), int(e["length"][i]), int(e["aux"][i])))
    sam.states = states
    if kind == 0:
        tok, dst, n = ora.export_topk()
        sam.states_topk_next = [list(zip(tok[s, :n[s]].tolist(), dst[s, :n[s]].tolist())) for s in range(len(states))]
        sam.max_predicts, sam.alpha, sam.device, sam.K = 60, 4.0, "cuda", 8
    else:
        sam.n_predicts = 40
        sam.input_ids = [int(t) for t in e["text"]]
    sam.last, sam.max_length, sam.cur_index, sam.cur_length = 7, 11, 3, 2
    return sam, ora


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("protocol", [2, 3, 4, 5])
def test_every_protocol_and_both_variants(tmp_path, monkeypatch, kind, protocol):
    rng = np.random.default_rng(10 * kind + protocol)
    docs = [markov_stream(rng, 150, vocab=60) for _ in range(8)] + [[i] for i in range(60)]          # the EOS state is a hub: spill blocks
    sam, ora = reference_like_object(kind, docs, 2, monkeypatch)
    path = str(tmp_path / "ref.pkl")
    with open(path, "wb") as f:
        pickle.dump(sam, f, protocol=protocol)
    auto, pickled = samd_hip.StaticAutomaton.from_reference_pickle(path, kind)
    built = samd_hip.StaticAutomaton.build(docs, 2, kind)
    same_tables(tables(auto), tables(built))
    assert (pickled["last"], pickled["max_length"], pickled["cur_index"], pickled["cur_length"]) == (7, 11, 3, 2)
    if kind == 0:
        assert (pickled["max_predicts"], pickled["alpha"], pickled["K"]) == (60, 4.0, 8)
    else:
        assert pickled["n_predicts"] == 40
        img = str(tmp_path / "a.samd")
        auto.save(img); built.save(str(tmp_path / "b.samd"))
        assert open(img, "rb").read() == open(str(tmp_path / "b.samd"), "rb").read()                # text region included
    # the other variant's reader declines it
    h = C.c_void_p()
    assert samd_hip.lib().samd_static_from_pickle(os.fsencode(path), 1 - kind, None, C.byref(h)) == -4


def test_nothing_is_executed_and_foreign_pickles_are_declined(tmp_path):
    marker = str(tmp_path / "executed")

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    cases = {"evil.pkl": pickle.dumps(Evil()), "dict.pkl": pickle.dumps({"states": [1, 2, 3]}), "garbage.pkl": b"\x80\x04\x95\xff" + os.urandom(64),
             "empty.pkl": b"", "text.pkl": pickle.dumps([1, 2, 3], protocol=0)}
    good = open(os.path.join(HERE, "golden", "ref_static_sam.pkl"), "rb").read()
    for cut in (1, 10, 100, len(good) // 2, len(good) - 1):
        cases[f"cut{cut}.pkl"] = good[:cut]
    for name, data in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        h = C.c_void_p()
        rc = samd_hip.lib().samd_static_from_pickle(os.fsencode(p), 0, None, C.byref(h))
        assert rc == -4 and not h.value, (name, rc)
        assert samd_hip.lib().samd_last_error()
    assert not os.path.exists(marker)
    # bit flips of a good pickle: declined or a structurally sound automaton, never a crash (ASan build: scripts/asan_cpu.sh)
    rng = np.random.default_rng(3)
    arr = np.frombuffer(good, dtype=np.uint8)
    for it in range(1500):
        m = arr.copy()
        i = int(rng.integers(0, len(m)))
        m[i] ^= np.uint8(1 << int(rng.integers(0, 8)))
        p = str(tmp_path / "flip.pkl")
        m.tofile(p)
        h = C.c_void_p()
        rc = samd_hip.lib().samd_static_from_pickle(os.fsencode(p), 0, None, C.byref(h))
        assert rc in (0, -1, -2, -4), (it, rc)
        if rc == 0:
            a = samd_hip.StaticAutomaton(h)
            e = a.export()
            n = a.info()["n_states"]
            assert ((e["link"][1:] >= 0) & (e["link"][1:] < n)).all() and ((e["edge_dst"] >= 0) & (e["edge_dst"] < n)).all()


def _declined(tmp_path, data, name="crafted.pkl"):
    p = str(tmp_path / name)
    open(p, "wb").write(data)
    h = C.c_void_p()
    rc = samd_hip.lib().samd_static_from_pickle(os.fsencode(p), 0, None, C.byref(h))
    assert rc == -4 and not h.value, (name, rc, data[:64])
    return samd_hip.lib().samd_last_error()


def test_a_memoized_container_cannot_be_reached_twice(tmp_path):
    """a box reached through BINPUT / BINGET could be freed by the discarding `states_topk_next` sink while a copy sat on the
    stack; BUILD then read items[0] / items[1] of the dead (or recycled) box.  The 40-byte stream the advisor crashed the library with,
    the same idea through LONG_BINPUT / LONG_BINGET / MEMOIZE, and a GET of a freed object after its box id was handed out again."""
    obj = b"cm\nStaticSAM\n)\x81"                                        # GLOBAL m.StaticSAM, EMPTY_TUPLE, NEWOBJ -> an object box
    key = b"\x8c\x10states_topk_next"                                     # SHORT_BINUNICODE 'states_topk_next': the next list discards its items
    streams = {
        "advisor": obj + b"q\x00" + key + b"]h\x00a00}b.",                # PUT 0, list, GET 0, APPEND (frees the object), POP, POP, BUILD on the dead box
        "long":    obj + b"r\x00\x00\x00\x00" + key + b"]j\x00\x00\x00\x00a00}b.",
        "memoize": b"\x80\x04" + obj + b"\x94" + key + b"]h\x00a00}b.",
        "reuse":   obj + b"q\x00" + key + b"]h\x00a00" + b"}" + b"h\x00" + b"}b.",   # ... a new dict takes the freed id, GET 0 again, BUILD
        "list_in_list": key + b"]q\x01" + key + b"]h\x01a0h\x01.",       # a discarding list appended to another one, then fetched again
        "tuple_get": b"K\x01K\x02\x86q\x05" + key + b"]h\x05a0h\x05.",
    }
    for name, data in streams.items():
        msg = _declined(tmp_path, data, name + ".pkl")
        assert msg
    # mutator: every BINGET / LONG_BINGET operand of the reference's own pickle re-pointed at every memo slot that holds a container
    good = open(os.path.join(HERE, "golden", "ref_static_sam.pkl"), "rb").read()
    import pickletools
    ops = list(pickletools.genops(good))
    gets = [(pos, op.name, arg) for op, arg, pos in ops if op.name in ("BINGET", "LONG_BINGET")]
    memo_ops = [(pos, op.name) for op, arg, pos in ops if op.name in ("MEMOIZE", "BINPUT", "LONG_BINPUT")]
    assert gets and memo_ops
    rng = np.random.default_rng(11)
    n_memo = len(memo_ops)
    tried = 0
    for pos, name, arg in gets[:: max(1, len(gets) // 40)]:
        width = 1 if name == "BINGET" else 4
        for target in {0, 1, 2, 3, int(rng.integers(0, min(n_memo, 256 ** width))), int(rng.integers(0, min(n_memo, 256 ** width)))}:
            if target == arg:
                continue
            m = bytearray(good)
            m[pos + 1:pos + 1 + width] = int(target).to_bytes(width, "little")
            p = str(tmp_path / "get.pkl")
            open(p, "wb").write(bytes(m))
            h = C.c_void_p()
            rc = samd_hip.lib().samd_static_from_pickle(os.fsencode(p), 0, None, C.byref(h))
            assert rc in (0, -1, -2, -4), (pos, target, rc)                # declined or sound -- never a crash (the ASan build runs this too)
            if rc == 0:
                samd_hip.StaticAutomaton(h).export()
            tried += 1
    assert tried >= 40


def test_load_sam_falls_back_to_pickle_load_with_a_warning(tmp_path, monkeypatch):
    """protocol 0/1 text pickles are outside the reader's subset: load_sam still loads them through the object graph"""
    docs = [[3, 4, 5, 3, 4, 6], [7, 3, 4]]
    sam, ora = reference_like_object(0, docs, 2, monkeypatch)
    path = str(tmp_path / "p1.pkl")
    with open(path, "wb") as f:
        pickle.dump(sam, f, protocol=1)
    monkeypatch.undo()                                              # the real samd_sam_only.sam.static_sam.StaticSAM resolves again
    import samd_sam_only
    with pytest.warns(RuntimeWarning, match="falling back to pickle.load"):
        got = samd_sam_only.load_sam(path)
    same_tables(tables(got._auto), tables(samd_hip.StaticAutomaton.build(docs, 2, 0)))


_SCALE_CHILD = r"""
import json, os, sys, time
sys.path[:0] = [sys.argv[3], os.path.join(sys.argv[3], "sam-decoding_amd")]
def hwm():
    for line in open("/proc/self/status"):
        if line.startswith("VmHWM"):
            return int(line.split()[1]) * 1024
import samd_hip, samd_sam_only
samd_hip.lib()
base = hwm()
t0 = time.perf_counter()
how = sys.argv[2]
if how == "native":
    sam = samd_sam_only.load_sam(sys.argv[1])
else:
    import pickle
    with open(sys.argv[1], "rb") as f:
        sam = pickle.load(f)
    sam.init_topk_next()
dt = time.perf_counter() - t0
info = sam._auto.info()
sam._auto.save(sys.argv[1] + "." + how + ".samd")
print(json.dumps({"how": how, "seconds": dt, "peak_rss_delta": hwm() - base, "n_states": info["n_states"], "image_bytes": info["device_bytes"]}))
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "samd_sam_only")), reason="needs the reference checkout (dev container only)")
@pytest.mark.skipif(bool(os.environ.get("SAMD_HIP_LIB")), reason="a timing / memory measurement: not under the sanitizer build")
def test_a_pickle_of_2_to_the_20_tokens_written_by_the_reference(tmp_path):
    """the reference builds (StaticSAM.build), pickles (dump_sam) -- this process only orchestrates; the import runs in a child so that its
    peak memory is its own.  Bars: state table == the oracle's; native peak RSS <= 3 x the image; reported beside the pickle.load route."""
    sys.path.insert(0, ROOT)
    import bench
    n_tok = int(os.environ.get("SAMD_PICKLE_SCALE_TOKENS", 1 << 20))
    flat, off, docs = bench.synth_corpus(n_tok, vocab=2048)         # (a small vocabulary keeps the reference's Python build at ~1 minute)
    pkl = str(tmp_path / "ref_big.pkl")
    gen = (
        "import sys, types, pickle, numpy as np\n"
        f"R = {REF!r}\n"
        "sys.path.insert(0, R)\n"
        "for pkg in ('samd_sam_only', 'samd'):\n"
        "    m = types.ModuleType(pkg); m.__path__ = [f'{R}/{pkg}']; sys.modules[pkg] = m\n"
        "from samd_sam_only.sam.static_sam import StaticSAM\n"
        "d = np.load(sys.argv[1])\n"
        "flat, off = d['flat'], d['off']\n"
        "docs = [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]\n"
        "sam = StaticSAM.build(docs, 2, verbose=False)\n"
        "with open(sys.argv[2], 'wb') as f:\n"
        "    pickle.dump(sam, f)\n"                                  # == dump_sam (SO/sam/utils.py:20-22); utils.py itself imports `datasets`
        "print(len(sam.states))\n")
    npz = str(tmp_path / "corpus.npz")
    np.savez(npz, flat=flat, off=off)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", gen, npz, pkl], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    n_ref = int(r.stdout.strip().splitlines()[-1])
    ref_build_s = time.perf_counter() - t0
    out = {}
    for how in ("native", "pickle"):
        c = subprocess.run([sys.executable, "-c", _SCALE_CHILD, pkl, how, ROOT], capture_output=True, text=True, timeout=1500)
        assert c.returncode == 0, c.stderr[-2000:]
        out[how] = json.loads(c.stdout.strip().splitlines()[-1])
        assert out[how]["n_states"] == n_ref
    assert open(pkl + ".native.samd", "rb").read() == open(pkl + ".pickle.samd", "rb").read()       # the same image either way
    st = O.StaticSAM()
    fa, fp = O._i32(flat)
    oa, op = O._i64(off)
    O.lib().osam_add_batch(st._h, fp, op, len(off) - 1, 2)
    st.init_topk_next()
    built = samd_hip.StaticAutomaton.load(pkl + ".native.samd").export()
    oe = st.export()
    for k in ("link", "length", "aux", "deg"):
        assert np.array_equal(built[k], oe[k]), k
    img = out["native"]["image_bytes"]
    report = {"tokens": n_tok, "states": n_ref, "pickle_bytes": os.path.getsize(pkl), "image_bytes": img, "reference_build_and_dump_s": round(ref_build_s, 1),
              "native": out["native"], "pickle_load": out["pickle"]}
    print("PICKLE_SCALE", json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "pickle_scale.json"), "w"), indent=1)
    assert out["native"]["peak_rss_delta"] <= 3 * img, report
    assert out["native"]["peak_rss_delta"] < out["pickle"]["peak_rss_delta"] and out["native"]["seconds"] < out["pickle"]["seconds"]
