"""Host-side pieces of the product that run without a GPU: the static-automaton builder / image
layout / save+load in libsamd_hip.so, and the C-ABI export list.  CPU only."""
import os
import re

import numpy as np
import pytest

import samd_hip
from oracle import sam_oracle as O
from util import markov_stream, split_edges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "samd_hip.h")).read()
    declared = set(re.findall(r"\b(samd_[a-z0-9_]+)\s*\(", hdr))
    L = samd_hip.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/samd_hip.h but not exported"
    assert declared == set(samd_hip._PROTOS), declared ^ set(samd_hip._PROTOS)


def check_against_oracle(docs, eos, kind):
    prod = samd_hip.StaticAutomaton.build(docs, eos, kind)
    ora = (O.StaticSAM if kind == 0 else O.StaticSAMFull).build(docs, eos)
    pt, ot = prod.export(), ora.export()
    for k in ("link", "length", "aux", "deg"):
        assert np.array_equal(pt[k], ot[k]), k
    pe, oe = split_edges(pt), split_edges(ot)
    for s, (a, b) in enumerate(zip(pe, oe)):
        assert sorted(a) == sorted(b), s
    if kind == 0:
        tok, dst, n = ora.export_topk()
        for s, a in enumerate(pe):
            assert a[:min(len(a), 8)] == list(zip(tok[s, :n[s]].tolist(), dst[s, :n[s]].tolist())), s
            assert [t for t, _ in a[8:]] == sorted(t for t, _ in a[8:])
    else:
        for s, (a, b) in enumerate(zip(pe, oe)):
            assert a[:8] == b[:8]            # dict order for the ENDPOS kind
    return prod, ora


def test_builder_matches_golden_and_oracle(golden):
    g = golden("sam_traces.json.gz")
    for case in g["static_so"]:
        prod, _ = check_against_oracle(case["docs"], case["eos"], 0)
        pe = split_edges(prod.export())
        for s, want in enumerate(case["topk"]):
            assert [list(x) for x in pe[s][:len(want)]] == want      # the reference's own top-k table
    for case in g["static_s"]:
        check_against_oracle(case["docs"], case["eos"], 1)


@pytest.mark.parametrize("vocab,n_docs,doc_len", [(6, 30, 60), (50, 40, 200), (3000, 20, 400)])
def test_builder_random_corpora(vocab, n_docs, doc_len):
    rng = np.random.default_rng(vocab)
    docs = [markov_stream(rng, doc_len, vocab=vocab) for _ in range(n_docs)] + [[i] for i in range(vocab)]
    for kind in (0, 1):
        check_against_oracle(docs, 2, kind)


def test_save_load_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    docs = [markov_stream(rng, 120, vocab=40) for _ in range(10)] + [[i] for i in range(40)]
    a = samd_hip.StaticAutomaton.build(docs, 2, 0)
    p = str(tmp_path / "sam.bin")
    a.save(p)
    b = samd_hip.StaticAutomaton.load(p)
    ea, eb = a.export(), b.export()
    for k in ea:
        assert np.array_equal(ea[k], eb[k])
    assert a.info() == b.info()
    with open(p, "r+b") as f:
        f.write(b"garbage!")
    with pytest.raises(samd_hip.SamdError):
        samd_hip.StaticAutomaton.load(p)


def test_from_tables_matches_build():
    rng = np.random.default_rng(9)
    docs = [markov_stream(rng, 150, vocab=30) for _ in range(8)] + [[i] for i in range(30)]
    ora = O.StaticSAM.build(docs, 2)
    t = ora.export()
    a = samd_hip.StaticAutomaton.from_tables(0, t["link"], t["length"], t["aux"], t["deg"], t["edge_tok"], t["edge_dst"])
    b = samd_hip.StaticAutomaton.build(docs, 2, 0)
    ea, eb = a.export(), b.export()
    for k in ea:
        assert np.array_equal(ea[k], eb[k]), k


def test_errors_are_statuses_not_crashes():
    with pytest.raises(samd_hip.SamdError):
        samd_hip.StaticAutomaton.build([[1, 2], []], 2, 0)          # empty document
    with pytest.raises(samd_hip.SamdError):
        samd_hip.StaticAutomaton.load("/nonexistent/sam.bin")


def test_no_gpu_means_loud_failure():
    if samd_hip.lib().samd_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(samd_hip.SamdError):
        samd_hip.Session(128)
    a = samd_hip.StaticAutomaton.build([[3, 4, 5]], 2, 0)
    with pytest.raises(samd_hip.SamdError):
        a.upload()


def test_load_rejects_damaged_image(tmp_path):
    """A SAMDHIP1 file whose indices point outside the image must be refused at load time (samd_static_load), not
    discovered by a kernel."""
    import struct
    rng = np.random.default_rng(5)
    docs = [rng.integers(0, 40, 60).tolist() for _ in range(3)]
    sam = samd_hip.StaticAutomaton.build(docs, 39, 0)
    good = tmp_path / "good.samd"
    sam.save(str(good))
    raw = bytearray(good.read_bytes())
    inf = sam.info()
    assert samd_hip.StaticAutomaton.load(str(good)).info()["n_states"] == inf["n_states"]
    # header = magic[8] + 6 x int64 ... find the first node by searching for state 0's link (-1) after the header
    hdr = len(raw) - (inf["n_states"] * 64 + inf["vocab"] * 4 + inf["n_spill"] * 8 + inf["n_text"] * 4)
    assert hdr > 0
    bad = bytearray(raw)
    struct.pack_into("<i", bad, hdr + 64 * 1 + 12, inf["n_states"] + 7)          # state 1: e0.dst past the end
    p = tmp_path / "bad_edge.samd"; p.write_bytes(bytes(bad))
    with pytest.raises(samd_hip.SamdError, match="damaged"):
        samd_hip.StaticAutomaton.load(str(p))
    bad = bytearray(raw)
    struct.pack_into("<i", bad, hdr + 64 * 2 + 0, -5)                          # state 2: suffix link < -1
    p = tmp_path / "bad_link.samd"; p.write_bytes(bytes(bad))
    with pytest.raises(samd_hip.SamdError, match="damaged"):
        samd_hip.StaticAutomaton.load(str(p))
    p = tmp_path / "short.samd"; p.write_bytes(bytes(raw[:-16]))
    with pytest.raises(samd_hip.SamdError, match="truncated"):
        samd_hip.StaticAutomaton.load(str(p))


def test_gemm_kernel_isa_keeps_the_hand_counted_waits_valid(tmp_path):
    """k_gemm_skinny waits with hand-counted s_waitcnt vmcnt(N) (two weight chunks in flight).  That count is only right
    while the compiler adds no vector-memory instruction of its own between the hand-issued loads: no scratch spills, and
    no wait other than vmcnt(0) / vmcnt(8 + XV) inside the kernels.  Checked on the generated gfx950 assembly."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "sam-decoding_amd", "csrc", "gemm_kernels.hip")
    out = str(tmp_path / "gemm.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I" + os.path.join(ROOT, "include"), "-o", out, src], stderr=subprocess.DEVNULL)
    asm = open(out).read()
    assert "scratch_" not in asm, "a register spill would add vector-memory instructions the wait counts do not know about"
    kernels = re.findall(r"^(_Z13k_gemm_skinny\w+):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) >= 12
    for name, body in kernels:
        rt = int(re.search(r"Li(\d)ELi\d", name).group(1))
        allowed = {0, 8 + rt}                                   # XV = rows * 32 / 512 = RT
        waits = {int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)}
        assert waits <= allowed, (name, waits)
        assert "global_load_lds_dwordx4" in body and "ds_read_b128" in body
    for m in re.finditer(r"\.name:\s+_Z13k_gemm_skinny.*?\.vgpr_spill_count:\s+(\d+)", asm, flags=re.S):
        assert int(m.group(1)) == 0
