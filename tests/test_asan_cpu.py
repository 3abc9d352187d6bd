"""CPU sanitizers: scripts/asan_cpu.sh builds libsamd_hip.so with AddressSanitizer + UBSan on the host pass and the
oracle likewise, then runs the builder / corpus CLI / oracle-vs-golden tests and the SAMDHIP1 loader fuzz against them.  This test
drives the script (shorter fuzz) and proves the instrumentation is live with a negative control: an export into arrays one element
short must abort with an AddressSanitizer report.  CPU only -- GPU sanitizers do not exist on this pool."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "asan_cpu.sh")

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("bash") is None, reason="needs the ROCm toolchain")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "SAMD_HIP_LIB", "SAM_ORACLE_LIB", "ASAN_OPTIONS", "UBSAN_OPTIONS")}
    e.update(kw)
    return e


def test_cpu_paths_are_clean_under_asan_and_ubsan():
    r = subprocess.run(["bash", SCRIPT], capture_output=True, text=True, timeout=1500, env=_env(SAMD_FUZZ_N="3000"), cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_negative_control_a_heap_overflow_is_caught():
    """samd_static_export into edge arrays ONE element short: the sanitizer build must abort (so a clean run above means something)"""
    code = (
        "import sys, ctypes as C, numpy as np\n"
        f"sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'sam-decoding_amd')!r}]\n"
        "import samd_hip\n"
        "a = samd_hip.StaticAutomaton.build([[3, 4, 5, 3, 4, 6, 7, 8, 9, 3]], 2, 0)\n"
        "i = a.info(); n, ne = i['n_states'], i['n_edges']\n"
        "link, length, aux, deg = (np.empty(n, np.int32) for _ in range(4))\n"
        "et, ed = np.empty(ne - 1, np.int32), np.empty(ne - 1, np.int32)\n"
        "P = samd_hip._ptr\n"
        "samd_hip.lib().samd_static_export(a._h, P(link), P(length), P(aux), P(deg), P(et), P(ed))\n"
        "print('NOT CAUGHT')\n")
    r = subprocess.run(["bash", SCRIPT, "--run", sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode != 0 and "NOT CAUGHT" not in r.stdout, r.stdout[-500:]
    assert "AddressSanitizer" in r.stderr and "heap-buffer-overflow" in r.stderr, r.stderr[-1500:]
