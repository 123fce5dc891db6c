#!/bin/bash
# usage: ab_variant.sh NAME ["EXTRA compiler flags"] [git-rev]
# Builds opal_amd/lib/libvar_NAME.so from the working tree (or from git-rev) in a scratch copy, with EXTRA
# appended to the device compile flags (e.g. "-DOPAL_PAIR_WAVES=5"), for same-box A/B runs (ab_run2.sh).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; REV=$3
TMP=$(mktemp -d)
if [ -n "$REV" ]; then
  git -C "$ROOT" archive "$REV" opal_amd/csrc include | tar -x -C "$TMP"
else
  mkdir -p "$TMP/opal_amd" && cp -r "$ROOT/opal_amd/csrc" "$TMP/opal_amd/" && cp -r "$ROOT/include" "$TMP/" && rm -f "$TMP"/opal_amd/csrc/*.o
fi
make -s -C "$TMP/opal_amd/csrc" -j8 EXTRA="$EXTRA" OUT="$ROOT/opal_amd/lib/libvar_$NAME.so" "$ROOT/opal_amd/lib/libvar_$NAME.so" 2>&1 | grep -E "error|Error" || true
rm -rf "$TMP"
ls -la "$ROOT/opal_amd/lib/libvar_$NAME.so"
