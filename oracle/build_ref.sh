#!/usr/bin/env bash
# build_ref.sh -- compile the REAL reference (`dw`) from the sources where they
# lie under /root/reference into oracle/_ref/dw (git-ignored, travels to the GPU
# box).  TEST INFRASTRUCTURE ONLY: nothing in sampler_amd/ may use it.
#
# Recipe: the reference's own `make` is NOT run.  g++ is invoked
# directly on the 14 files of the reference Makefile's SOURCES list
# (/root/reference/Makefile:41-54).  The two vendored header-only dependencies the
# sources include (TCLAP 1.2.1 for cmd_parser.cc, gtest_prod.h for common.h) are
# unpacked from the reference's own lib/*.tar.gz|zip into a scratch directory
# under $TMPDIR -- never into this repository.  Two accommodations for g++ 11:
# force-include <functional>/<memory> (missing includes in text2bin.cc/common.h)
# and no -Werror.
set -euo pipefail
REF=${DW_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
if [[ ! -d "$REF/src" ]]; then
  echo "build_ref.sh: $REF not present; keeping prebuilt $OUT (if any)" >&2
  exit 0
fi
SCRATCH=$(mktemp -d "${TMPDIR:-/tmp}/dw_refbuild.XXXXXX")
trap 'rm -rf "$SCRATCH"' EXIT
tar -xzf "$REF/lib/tclap-1.2.1.tar.gz" -C "$SCRATCH"
python3 - "$REF/lib/gtest-1.7.0.zip" "$SCRATCH" <<'PY'
import sys, zipfile
zipfile.ZipFile(sys.argv[1]).extractall(sys.argv[2])
PY
SRC=(dimmwitted.cc cmd_parser.cc binary_format.cc bin2text.cc text2bin.cc
     main.cc weight.cc variable.cc factor.cc factor_graph.cc inference_result.cc
     gibbs_sampler.cc timer.cc numa_nodes.cc)
mkdir -p "$OUT"
g++ -include functional -include memory -std=c++11 -Wall -fno-strict-aliasing -Ofast \
    -I"$SCRATCH/tclap-1.2.1/include" -I"$SCRATCH/gtest-1.7.0/include" \
    -o "$OUT/dw" "${SRC[@]/#/$REF/src/}" -lnuma -lrt -lpthread 2>"$OUT/build.log" \
  || { cat "$OUT/build.log" >&2; exit 1; }
echo "built $OUT/dw"
