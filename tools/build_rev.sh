#!/bin/bash
# The library of another REVISION as an A/B candidate: tools/build_rev.sh <git-rev> <name>  ->  build/variants/<name>/libvgpu.so
# (a scratch worktree under /tmp, built with valida_amd/build.py, removed again; candidates of tools/gpu_session.sh: <name>=LIB:build/variants/<name>/libvgpu.so)
set -eu
REV=${1:?revision}; NAME=${2:?name}; ROOT=$(git rev-parse --show-toplevel); T=$(mktemp -d /tmp/vgpu_rev.XXXXXX)
git -C "$ROOT" worktree add -f "$T" "$REV" -q
( cd "$T" && python -m valida_amd.build > /dev/null )
mkdir -p "$ROOT/build/variants/$NAME" && cp "$T/valida_amd/libvgpu.so" "$ROOT/build/variants/$NAME/libvgpu.so"
git -C "$ROOT" worktree remove --force "$T"; git -C "$ROOT" worktree prune
echo "$ROOT/build/variants/$NAME/libvgpu.so  ($(git -C "$ROOT" rev-parse --short "$REV"))"
