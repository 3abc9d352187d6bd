"""Static checks on the compiled gfx950 code objects inside libsamd_hip.so (no GPU needed).

The streaming GEMM kernels issue their loads by hand (`asm volatile("global_load_dwordx4 ...")`) and wait with counted `s_waitcnt vmcnt(N)`.
A destination register the compiler SPILLS is stored to scratch right after the load was ISSUED, i.e. before the data has landed: the kernel
then computes on stale registers without any error (met once: profiles/r03_norm_fold.md).  So: no kernel of gemm_kernels.hip may spill a
vector register or use scratch."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "sam-decoding_amd", "samd_hip", "libsamd_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def gfx950_code_objects(blob):
    """every gfx950 ELF of the clang offload bundles embedded in the shared library"""
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            return
        pos = i + len(magic)
        n = struct.unpack_from("<Q", blob, i + 24)[0]
        off = i + 32
        if n > 16:
            continue
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple and size:
                yield blob[i + o:i + o + size]


@pytest.mark.skipif(not (os.path.exists(SO) and os.path.exists(READELF)), reason="needs the built library and llvm-readelf")
def test_hand_issued_load_kernels_do_not_spill(tmp_path):
    blob = open(SO, "rb").read()
    kernels = {}
    for k, co in enumerate(gfx950_code_objects(blob)):
        path = tmp_path / f"co{k}.elf"
        path.write_bytes(co)
        notes = subprocess.run([READELF, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
        for block in notes.split(".name:")[1:]:
            name = block.split()[0]
            get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
            kernels[name] = dict(scratch=get("private_segment_fixed_size"), vgpr_spills=get("vgpr_spill_count"), vgprs=get("vgpr_count"))
    gemm = {n: v for n, v in kernels.items() if "k_gemm_" in n and "pack" not in n}
    assert len(gemm) >= 20, f"expected the streaming GEMM instantiations, found {sorted(gemm)}"
    bad = {n: v for n, v in gemm.items() if v["scratch"] or v["vgpr_spills"]}
    assert not bad, f"kernels with hand-issued loads must not spill: {bad}"
