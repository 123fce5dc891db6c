#!/bin/bash
# Builds opal_amd/lib/libvar_${2:-base}.so from the committed HEAD sources (for A/B runs against the working tree).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive ${1:-HEAD} opal_amd/csrc include | tar -x -C "$TMP"
make -s -C "$TMP/opal_amd/csrc" -j8 OUT="$ROOT/opal_amd/lib/libvar_${2:-base}.so" "$ROOT/opal_amd/lib/libvar_${2:-base}.so" >/dev/null 2>&1 || make -C "$TMP/opal_amd/csrc" -j8 OUT="$ROOT/opal_amd/lib/libvar_${2:-base}.so" "$ROOT/opal_amd/lib/libvar_${2:-base}.so"
rm -rf "$TMP"
ls -la "$ROOT/opal_amd/lib/"
