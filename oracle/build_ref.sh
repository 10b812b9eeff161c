#!/bin/bash
# Build the REFERENCE's own OpenCL kernels, unmodified, for gfx950 with the ROCm OpenCL
# toolchain that ships in this image (clang -x cl + the ROCm device library: real
# get_global_id / max / select / vload4 -- nothing is stubbed).
#
# Sources are read where they lie under $NGM_REFERENCE (default /root/reference); nothing is
# copied into the repo.  Outputs go to oracle/_ref/ only (git-ignored, travels to the GPU box).
# The -D options are the ones NGM's host passes at JIT time
# (lib/mason/opencl/SWOcl.cpp:206-242, lib/mason/opencl/SWOclCigar.cpp:35).
#
# usage: build_ref.sh <gpu|gpu1|cpu> <qry_max_len> <corridor> [match mismatch gap_read gap_ref [bs|slam match_bonus_tt match_bonus_tc]]
#   (penalties positive, as in NGM's config; they are negated exactly as SWOcl.cpp:211-213 does)
#   bs / slam: the -D__ALT_SCORING__ builds of `--bs-mapping` / `--slam-seq 2` (SWOcl.cpp:225-242); their kernels take one more
#   argument, the per-pair `direction` bytes
set -euo pipefail
variant=${1:?variant gpu|cpu}; q=${2:?qry_max_len}; c=${3:?corridor}
match=${4:-10}; mismatch=${5:-15}; gap_read=${6:-20}; gap_ref=${7:-20}
alt=${8:-none}; tt=${9:-0}; tc=${10:-0}
REF=${NGM_REFERENCE:-/root/reference}
SRC=$REF/lib/mason/opencl/opencl
here=$(cd "$(dirname "$0")" && pwd)
out=$here/_ref
mkdir -p "$out"
[ -d "$SRC" ] || { echo "reference sources not found at $SRC" >&2; exit 3; }

il=256
if [ "$variant" = gpu ]; then
	tpb=256; vdef=-D__GPU__     # SWOcl.h:60-64, SWOcl.cpp:219-221
elif [ "$variant" = gpu1 ]; then
	# same __GPU__ kernels with one work-item per group (the value SWOcl.h:60 uses on __APPLE__):
	# sidesteps the racy sentinel store of oclSW_ScoreGlobal (oclEndFreeScore.cl:229 lacks
	# "* threads_per_block", so with 256 work-items it clobbers a neighbour's H[0])
	tpb=1; il=1; vdef=-D__GPU__
else
	tpb=1;   vdef=-D__CPU__     # float4 lanes, 4 pairs per work-item; run with local size 1
fi
name=ngm_ocl_${variant}_q${q}_c${c}_m${match}_x${mismatch}_gr${gap_read}_gf${gap_ref}
altdef="-DmatchALT=0 -DmismatchALT=0 -DscoresFWD=scores -DscoresREV=scores"
if [ "$alt" = bs ]; then
	altdef="-D__ALT_SCORING__ -DmatchALT=$tt -DmismatchALT=$tc -DscoresFWD=scoresBsFWD -DscoresREV=scoresBsREV"; name=${name}_bs_tt${tt}_tc${tc}
elif [ "$alt" = slam ]; then
	altdef="-D__ALT_SCORING__ -DmatchALT=$tt -DmismatchALT=-$tc -DscoresFWD=scoresSlamSeqFWD -DscoresREV=scoresSlamSeqREV"; name=${name}_slam_tt${tt}_tc${tc}
fi
# program text = oclDefines + oclSwScore + oclEndFreeScore + oclSwCigar (SWOcl.cpp:250-251, SWOclCigar.cpp:33-36)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
cat "$SRC/oclDefines.cl" "$SRC/oclSwScore.cl" "$SRC/oclEndFreeScore.cl" "$SRC/oclSwCigar.cl" > "$tmp/program.cl"
/opt/rocm/lib/llvm/bin/clang -x cl -cl-std=CL1.2 -Xclang -finclude-default-header \
	-target amdgcn-amd-amdhsa -mcpu=gfx950 --rocm-path=/opt/rocm -O2 -w \
	-DMATRIX_LENGTH=$(( (c + 1) * tpb )) -Dinterleave_number=$il -Dthreads_per_block=$tpb \
	-Dmatch=$match -Dmismatch=-$mismatch -Dgap_read=-$gap_read -Dgap_ref=-$gap_ref \
	-Dread_length=$q -Dref_length=$(( q + c )) -Dcorridor_length=$(( c + 1 )) \
	-Dalignment_length=$(( 2 * q + c + 1 )) \
	-Dresult_number=4 -DCIGAR_M=0 -DCIGAR_I=1 -DCIGAR_D=2 -DCIGAR_N=3 -DCIGAR_S=4 -DCIGAR_H=5 \
	-DCIGAR_P=6 -DCIGAR_EQ=7 -DCIGAR_X=8 $vdef \
	$altdef \
	"$tmp/program.cl" -o "$out/$name.co"
echo "$out/$name.co"
