#!/bin/sh
# Compiles the reference's OWN bench harness, /root/reference/futhark/main.c, UNMODIFIED and
# from where it lies, against include/ray.h + libray_mi355x.so -- the drop-in proof for the
# Futhark-shaped boundary.  Output goes to oracle/_ref/ only (git-ignored, but it travels to
# the GPU box with the snapshot, where /root/reference does not exist).
#
# The reference's render path itself (Futhark) cannot be compiled here: there is no futhark
# compiler and the generated ray.c is git-ignored upstream (futhark/.gitignore:1-2).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF=/root/reference/futhark/main.c
if [ ! -f "$REF" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) -- keeping any prebuilt oracle/_ref"
  exit 0
fi
mkdir -p "$ROOT/oracle/_ref"
${CC:-cc} -O3 -std=gnu99 -I"$ROOT/include" -o "$ROOT/oracle/_ref/futhark_main" "$REF" \
  -L"$ROOT/raytracers_amd" -lray_mi355x -Wl,-rpath,'$ORIGIN/../../raytracers_amd' -lm
echo "built oracle/_ref/futhark_main from $REF"
