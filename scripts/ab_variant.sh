#!/bin/bash
# Builds opal_amd/lib/libvar_<name>.so from the working tree with extra compiler flags: scripts/ab_variant.sh ppt4 -DOPAL_MAX_PPT=4
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
cp -r "$ROOT/opal_amd/csrc" "$TMP/csrc"; cp -r "$ROOT/include" "$TMP/include"
mkdir -p "$TMP/opal_amd" && mv "$TMP/csrc" "$TMP/opal_amd/csrc"
rm -f "$TMP"/opal_amd/csrc/*.o
make -s -C "$TMP/opal_amd/csrc" -j8 EXTRA="$*" OUT="$ROOT/opal_amd/lib/libvar_$NAME.so" "$ROOT/opal_amd/lib/libvar_$NAME.so" >/dev/null 2>&1 || make -C "$TMP/opal_amd/csrc" -j8 EXTRA="$*" OUT="$ROOT/opal_amd/lib/libvar_$NAME.so" "$ROOT/opal_amd/lib/libvar_$NAME.so"
rm -rf "$TMP"
ls -la "$ROOT/opal_amd/lib/"
