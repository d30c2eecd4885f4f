#!/bin/bash
# The host-wave emulator of the kernel bodies (tests/emu) under AddressSanitizer (development tool): builds tests/emu/libemu_wave.so
# with -fsanitize=address and calls the emulator tests directly (pytest's own allocations upset the preloaded runtime).  Reads
# outside a buffer that a GPU run survives -- they land inside the allocation -- show up here.  Removes the instrumented library afterwards.
set -e
cd "$(dirname "$0")/.."
( cd tests/emu && g++ -std=c++20 -O1 -g -fsanitize=address -fno-omit-frame-pointer -fPIC -shared -pthread -ffp-contract=off -o libemu_wave.so emu_wave.cpp )
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:use_sigaltstack=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python - <<'PY' 2>&1 | grep -v "ASan doesn't fully support makecontext"
import sys
sys.path[:0] = ["tests", ".", "oracle"]
import test_emu_front, test_emu_wave
for mod in (test_emu_front, test_emu_wave):
    for name in sorted(dir(mod)):
        if not name.startswith("test_"):
            continue
        f = getattr(mod, name)
        marks = [m for m in getattr(f, "pytestmark", []) if m.name == "parametrize"]
        cases = [{}]
        for m in marks:  # stacked parametrize marks: the product of their value lists
            names = [x.strip() for x in m.args[0].split(",")]
            cases = [dict(c, **dict(zip(names, v if len(names) > 1 else (v,)))) for c in cases for v in m.args[1]]
        for c in cases:
            f(**c)
        print(name, "ok", flush=True)
PY
rm -f tests/emu/libemu_wave.so
