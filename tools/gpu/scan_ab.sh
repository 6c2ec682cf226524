#!/bin/bash
# A/B of the matcher's scan kernel: variants built into misc3d_amd/lib/<name> (tools/gpu/scan_ab.sh build "<name> <defs>" ...),
# on the GPU box each variant's C4 call under rocprofv3 --kernel-trace: the scan launches' durations, the call's wall clock, the
# pairs' checksum.   run: tools/gpu/scan_ab.sh run <name> ...   ("default" = the product library)
cd "$(dirname "$0")/../.."
if [ "$1" = "build" ]; then
    shift
    for spec in "$@"; do
        name=${spec%% *}; defs=${spec#* }
        rm -rf misc3d_amd/lib/obj_$name; cp -rp misc3d_amd/lib/obj misc3d_amd/lib/obj_$name; rm -f misc3d_amd/lib/obj_$name/m3d_match_mfma.hip.o
        make -C misc3d_amd/csrc lib -j8 DEFS="$defs" OBJDIR=../lib/obj_$name LIBDIR=../lib/$name 2>&1 | grep -E "error|Error"
        ls -la misc3d_amd/lib/$name/libmisc3d_amd.so
    done
    exit 0
fi
shift
export TMPDIR=/tmp
for name in "$@"; do
    v=$name; [ "$name" = default ] && v=""
    echo "=== $name"
    M3D_LIB_VARIANT=$v python tools/time_match.py | tail -2
    rm -rf /tmp/sab; M3D_LIB_VARIANT=$v M3D_MATCH_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sab -o m -- python tools/time_match.py > /dev/null 2>&1
    python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/sab/**/m_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if 'nn16' in k: print(f"   {k:60s} n={len(v):3d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
PY
done
