#!/bin/bash
# build everything from the repo root; prints the compiler errors and fails loudly (a piped `python __graft_entry__.py | tail` hides them)
cd "$(dirname "$0")/.." || exit 1
if python __graft_entry__.py > /tmp/ddp_build.log 2>&1; then tail -1 /tmp/ddp_build.log | cut -c1-120; else grep -E "error|failed" -A6 /tmp/ddp_build.log | head -60; echo "BUILD FAILED"; exit 1; fi
