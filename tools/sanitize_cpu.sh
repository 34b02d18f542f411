#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host-side C / C++ of the repository, on the CPU (the GPU pool
# offers no sanitizer): the oracle (oracle/fspt_oracle.c) under its stage tests, and the native scene builder
# (fspt_amd/csrc/scene_build.cpp: OBJ / MTL parsing, SAH BVH build, materials, auto-focus) under the tests that pin it to
# the reference's JS.  The rest of libfspt needs a GPU; its entry points are stubs in the builder's sanitizer library.
#   usage: bash tools/sanitize_cpu.sh        (from the repository root; builds into /tmp/fspt_san)
set -e
cd "$(dirname "$0")/.."
D=/tmp/fspt_san; mkdir -p $D
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared"
gcc $SAN -std=c11 -ffp-contract=off -fno-fast-math -mfma -fopenmp -o $D/liboracle_san.so oracle/fspt_oracle.c -lm
python3 - <<'PY'
import re, sys
sys.path.insert(0, ".")
from fspt_amd import _lib as L
src = open("fspt_amd/csrc/scene_build.cpp").read()
defined = set(re.findall(r"^int (fspt_\w+)\(", src, re.M))
with open("/tmp/fspt_san/stubs.cpp", "w") as f:
    f.write('#include <cstdarg>\n#include <cstdio>\nstatic char g_err[512];\n'
            'void fspt_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }\nextern "C" {\n')
    for n in L.SIGNATURES:
        if n in defined:
            continue
        f.write('const char *fspt_last_error(void) { return g_err; }\n' if n == "fspt_last_error"
                else f'int fspt_abi_version(void) {{ return {L.ABI_VERSION}; }}\n' if n == "fspt_abi_version"  # (the binding checks it at load)
                else f'int {n}(...) {{ fspt_set_error("{n}: not in the sanitizer build"); return -100; }}\n')
    f.write("}\n")
PY
g++ $SAN -std=c++17 -o $D/libfspt_san.so fspt_amd/csrc/scene_build.cpp $D/stubs.cpp
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
FSPT_ORACLE_LIB=$D/liboracle_san.so python3 -m pytest tests/test_goldens.py -q -p no:cacheprovider \
  -k "d2 or d3 or d6 or sample or bounce or camera or bvh_test or rnd_replay"
FSPT_LIB=$D/libfspt_san.so python3 -m pytest tests/test_goldens.py -q -p no:cacheprovider \
  -k "d0_native or mtl_parser or 70k_scene or 1M_scene or scene_file_loader_matches"
FSPT_LIB=$D/libfspt_san.so python3 tools/fuzz_builder.py
