#!/bin/bash
# oracle/build_dropin.sh -- TEST INFRASTRUCTURE: the REAL NextGenMap program with this repository's library plugged in
# behind IAlignment, to prove "drop-in" rather than "driven like" (VERDICT r1, item 8).
#   scratch copy of the reference's src/NGM.cpp  ->  oracle/dropin_patch.py (_NGM::CreateAlignment
#   calls the plugin exports)  ->  compiled with the flags of oracle/ngm_ref.mk  ->  linked with the UNCHANGED objects of
#   oracle/_ref/ngm/ngm-core (every other translation unit of the reference) and nextgenmap_amd/libngm_hip.so.
# Output: oracle/_ref/dropin/ngm-core-hip (git-ignored, travels to the GPU box).  Nothing of the reference is committed; the
# scratch copy lives in a temporary directory and is deleted.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(dirname "$HERE")
R=${NGM_REFERENCE:-/root/reference}
[ -d "$R/src" ] || { echo "no reference tree at $R: keeping the prebuilt oracle/_ref/dropin/"; exit 0; }
[ -f "$REPO/nextgenmap_amd/libngm_hip.so" ] || { echo "build nextgenmap_amd/libngm_hip.so first"; exit 1; }
make -s -C "$HERE" -f ngm_ref.mk NGM_REFERENCE="$R" "$HERE/_ref/ngm/ngm-core"
OUT="$HERE/_ref/dropin"; NGM="$HERE/_ref/ngm"
mkdir -p "$OUT"
SCRATCH=$(mktemp -d)
trap 'rm -rf "$SCRATCH"' EXIT
cp "$R/src/NGM.cpp" "$SCRATCH/NGM.cpp"
python3 "$HERE/dropin_patch.py" "$SCRATCH/NGM.cpp"
INC="-I$R/lib/seqan-library-1.4.1/include -I$R/lib/bamtools-2.3.0/src -I$R/lib/mason/opencl -I$R/include -I$R/src/parser -I$R/src/writer -I$R/src/core -I$R/src/misc -I$R/src/log -I$R/src/config -I$R/src -I$R/src/seqan -I$NGM/gen"
g++ -std=gnu++11 -fpermissive -w -D_BAM -pthread -O2 -DNDEBUG $INC -c "$SCRATCH/NGM.cpp" -o "$OUT/NGM_hip.o"
OBJS=$(find "$NGM/rel" "$NGM/bam" -name '*.o' ! -path "$NGM/rel/src/NGM.o" | sort)
g++ -pthread -o "$OUT/ngm-core-hip" $OBJS "$OUT/NGM_hip.o" -L"$REPO/nextgenmap_amd" -lngm_hip \
    -Wl,-rpath,'$ORIGIN/../../../nextgenmap_amd' -Wl,-rpath,/opt/rocm/lib -lz -lOpenCL
rm -f "$OUT/NGM_hip.o"
echo "$OUT/ngm-core-hip"
