#!/bin/sh
# Compile the reference's database-search library from its own two source files
# (opal.cpp + opal.h), in place, into oracle/_ref/. g++ only: the reference's
# 32-bit NW path has undefined behaviour that hangs clang builds.
# The five exported symbols are renamed ref_* so the library can be loaded next
# to libopal.so. Also builds the reference's own self-test and CLI *unchanged*
# and links them against OUR libopal.so: the drop-in check.
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
mkdir -p "$OUT"
CXXFLAGS="-std=c++11 -O3 -mavx2"
g++ $CXXFLAGS -fPIC -shared \
    -DopalSearchDatabase=ref_opalSearchDatabase \
    -DopalSearchDatabaseCharSW=ref_opalSearchDatabaseCharSW \
    -DopalInitSearchResult=ref_opalInitSearchResult \
    -DopalSearchResultIsEmpty=ref_opalSearchResultIsEmpty \
    -DopalSearchResultSetScore=ref_opalSearchResultSetScore \
    "$REF/src/opal.cpp" -o "$OUT/libopal_ref.so"
# Multi-threaded timing harness around the reference library (bench.py cpu_baseline leg).
gcc -O2 -fPIC -shared "$HERE/ref_bench.c" -L"$OUT" -lopal_ref -lpthread -Wl,-rpath,'$ORIGIN' -o "$OUT/libref_bench.so"
# Reference applications, reference objects only (for golden generation / comparison).
g++ $CXXFLAGS "$REF/src/opal.cpp" "$REF/src/ScoreMatrix.cpp" "$REF/src/opal_aligner.cpp" -o "$OUT/ref_opal_aligner"
g++ $CXXFLAGS "$REF/src/opal.cpp" "$REF/src/ScoreMatrix.cpp" "$REF/src/test.cpp" -o "$OUT/ref_test"
# Drop-in check: the reference's callers, unchanged, against our library.
LIB=$HERE/../opal_amd/csrc
if [ -f "$LIB/libopal.so" ]; then
    g++ $CXXFLAGS "$REF/src/ScoreMatrix.cpp" "$REF/src/opal_aligner.cpp" -L"$LIB" -lopal \
        -Wl,-rpath,'$ORIGIN/../../opal_amd/csrc' -o "$OUT/dropin_opal_aligner"
    g++ $CXXFLAGS "$REF/src/ScoreMatrix.cpp" "$REF/src/test.cpp" -L"$LIB" -lopal \
        -Wl,-rpath,'$ORIGIN/../../opal_amd/csrc' -o "$OUT/dropin_test"
else
    echo "build_ref.sh: $LIB/libopal.so not built yet; skipping drop-in binaries"
fi
echo "built: $(ls "$OUT")"
