#!/bin/bash
# One command on a machine that HAS the reference's dependencies (Eigen + Open3D 0.15.1; this repository's image has
# neither): builds tools/pin_reference/pin_reference.cpp against the real yuecideng/Misc3D headers with the
# reference's own flags (CMakeLists.txt:4,7,16: C++17, Release, -O3, no -march, no fast-math) and tells which
# floating-point association (M3D_FP_ORDER) reproduces it.
#   tools/pin_reference/run.sh <Misc3D checkout> [Open3D install prefix]      (Open3D_ROOT / CMAKE_PREFIX_PATH also work)
set -euo pipefail
REF=${1:?usage: run.sh <Misc3D checkout> [Open3D prefix]}
O3D=${2:-${Open3D_ROOT:-/usr/local}}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${TMPDIR:-/tmp}/m3d_pin
mkdir -p "$OUT"
make -C "$ROOT/oracle" -s
python3 "$HERE/pin.py" export "$OUT"
EIGEN_INC=$(ls -d "$O3D"/include/open3d/3rdparty 2>/dev/null || true)   # Open3D installs the Eigen it was built with here
# pin_seed.h is force-included in front of every translation unit that sees misc3d/utils.h: the reference's sampler then
# draws from std::mt19937(seed) instead of std::random_device (no reference source is modified)
# PIN_REGISTRATION=1 adds the Open3D RANSAC leg (Open3D 0.15.x: seed argument; newer: add -DPIN_O3D_GLOBAL_SEED)
g++ -std=c++17 -O3 -fopenmp -include "$HERE/pin_seed.h" ${PIN_REGISTRATION:+-DPIN_WITH_REGISTRATION} ${PIN_EXTRA_FLAGS:-} \
    -I"$HERE" -I"$REF/include" -I"$O3D/include" ${EIGEN_INC:+-I"$EIGEN_INC"} -I/usr/include/eigen3 \
    "$HERE/pin_reference.cpp" "$REF/src/logging.cpp" "$REF/src/iterative_plane_segmentation.cpp" \
    -L"$O3D/lib" -lOpen3D -Wl,-rpath,"$O3D/lib" -o "$OUT/pin_reference"
OMP_NUM_THREADS=1 "$OUT/pin_reference" "$OUT"      # one thread: the reference's loop is then the sequential one
python3 "$HERE/pin.py" compare "$OUT"
