#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer run of everything that executes on the CPU:
#   * the native static-automaton builder, the SAMDHIP1 image parser + structural check, export / host-image entry points
#     (sam-decoding_amd/csrc/sam_build.cpp and the host halves of the other sources), and
#   * the CPU oracle (oracle/sam_oracle.c),
# under tests/test_builder_cpu.py, test_gen_sam_cpu.py, test_oracle_golden.py, the loader fuzz tests/test_image_fuzz_cpu.py and the pickle reader tests/test_pickle_import_cpu.py.
#
# The library is the REAL one: every source compiled by hipcc with the sanitizers on the HOST pass only (-Xarch_host; the gfx950 code
# objects are the usual ones -- GPU sanitizers do not exist on this pool), so the tests load it through the ordinary binding
# (SAMD_HIP_LIB) and every declared symbol resolves.  The oracle is built with the same clang so that ONE sanitizer runtime
# (libclang_rt.asan, preloaded into python) serves both.  CPU only: nothing here needs or touches a GPU.
#
#   scripts/asan_cpu.sh [pytest args...]        exit code = pytest's; build products under build/asan/ (git-ignored)
#   scripts/asan_cpu.sh --run <command...>      the command under the same environment (preloaded runtime, SAMD_HIP_LIB, SAM_ORACLE_LIB)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/build/asan"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLANG="${CLANG:-/opt/rocm/lib/llvm/bin/clang}"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
XSAN=""
for f in $SAN; do XSAN="$XSAN -Xarch_host $f"; done
CSRC="$ROOT/sam-decoding_amd/csrc"
SRCS="sam_build.cpp sam_pickle.cpp sam_kernels.hip verify_kernels.hip attn_kernels.hip eagle_kernels.hip lm_kernels.hip gemm_kernels.hip"
LIB="$OUT/libsamd_hip.so"
newest=$(ls -t "$CSRC"/*.cpp "$CSRC"/*.hip "$CSRC"/*.h "$ROOT"/include/*.h | head -1)
if [ ! -f "$LIB" ] || [ "$newest" -nt "$LIB" ]; then
  echo "[asan] building $LIB (host pass with $SAN)"
  (cd "$CSRC" && "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off $XSAN -shared-libsan -o "$LIB" $SRCS)
fi
ORA="$OUT/libsam_oracle.so"
if [ ! -f "$ORA" ] || [ "$ROOT/oracle/sam_oracle.c" -nt "$ORA" ]; then
  echo "[asan] building $ORA"
  "$CLANG" -O1 -g -std=c99 -fPIC -Wall -Wextra $SAN -shared-libsan -shared -o "$ORA" "$ROOT/oracle/sam_oracle.c"
fi
RT="$("$CLANG" -print-file-name=libclang_rt.asan-x86_64.so)"
[ -f "$RT" ] || { echo "[asan] sanitizer runtime not found: $RT" >&2; exit 3; }
# (through a file, not `nm | grep -q`: with pipefail, grep leaving at its first match kills nm with SIGPIPE and the check fails at random)
nm -D "$LIB" > "$OUT/libsamd_hip.syms"; grep -q __asan_init "$OUT/libsamd_hip.syms" || { echo "[asan] $LIB is not instrumented" >&2; exit 3; }
nm -D "$ORA" > "$OUT/libsam_oracle.syms"; grep -q __asan_init "$OUT/libsam_oracle.syms" || { echo "[asan] $ORA is not instrumented" >&2; exit 3; }
cd "$ROOT"
# detect_leaks=0: the interpreter itself leaks by design at exit; everything else (heap / stack / global overflows, use after free,
# UB such as signed overflow, misaligned or null access, out-of-range shifts) aborts the run with a report
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=0:handle_segv=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}"
export LD_LIBRARY_PATH="$(dirname "$RT")${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
export SAMD_HIP_LIB="$LIB" SAM_ORACLE_LIB="$ORA"
if [ "${1:-}" = "--run" ]; then          # any command under the sanitizer environment (tests/test_asan_cpu.py: the negative control)
  shift
  exec "$@"
fi
if [ $# -eq 0 ]; then
  set -- tests/test_builder_cpu.py tests/test_gen_sam_cpu.py tests/test_oracle_golden.py tests/test_image_fuzz_cpu.py tests/test_pickle_import_cpu.py -x -q -p no:cacheprovider
fi
echo "[asan] python -m pytest $*"
exec python -m pytest "$@"
