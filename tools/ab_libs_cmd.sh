#!/bin/bash
# Dev tool: run one command under several builds of libudapose_hip.so (tools/_ab/lib_<tag>.so), interleaved.
# usage: tools/ab_libs_cmd.sh "<tag> <tag> ..." rounds <command ...>
TAGS="$1"; R="$2"; shift 2
L=uda_poseestimation_amd/libudapose_hip.so
cp $L /tmp/lib_keep.so
# (the production library is restored however the script ends - interrupt, timeout kill, failed cp)
trap 'cp /tmp/lib_keep.so $L' EXIT
for i in $(seq 1 $R); do
  for tag in $TAGS; do
    cp tools/_ab/lib_$tag.so $L
    echo "== $tag"; "$@" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
cp /tmp/lib_keep.so $L
