#!/bin/bash
# Where does one m3d_match_mutual_nn call (C4: 200 k x 200 k x 33) spend its wall clock?  Builds the library with
# -DM3D_MATCH_TIMELINE into misc3d_amd/lib/timeline (where hipcc is; the .so travels with gpurun); the call then prints the
# host's clock at its stations on stderr:
#   tools/match_timeline.sh build          (here)
#   tools/match_timeline.sh run 2> out.txt (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
    make -C misc3d_amd/csrc lib -j8 DEFS=-DM3D_MATCH_TIMELINE OBJDIR=../lib/obj_timeline LIBDIR=../lib/timeline > /dev/null
    ls -la misc3d_amd/lib/timeline/libmisc3d_amd.so
else
    M3D_LIB_VARIANT=timeline python tools/time_match.py
fi
