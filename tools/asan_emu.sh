#!/bin/bash
# AddressSanitizer pass over the kernels: the same csrc/ sources, host-emulated (tools/hipemu), every device buffer a
# heap allocation, so an out-of-range load/store in a kernel is reported with file:line.  (GPU ASan / XNACK are not
# available on the MI355X pool; a GPU memory fault only says "Aborted".)
#   tools/asan_emu.sh [pytest args]      default: tests/test_emu_parity.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/dif-pan_amd" -j8 emu-asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
DDIF_EMU_LIB="$ROOT/dif-pan_amd/lib/libddif_emu_asan.so" LD_PRELOAD="$RT" \
  ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  python -m pytest "${@:-tests/test_emu_parity.py}" -x -q
