cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for v in 0 1 auto 0 1; do
if [ $v = auto ]; then unset DDP_PEND_CHUNK; else export DDP_PEND_CHUNK=$v; fi
echo "chunk=$v $(python profiles/ilqg_queue_c3.py 2>&1 | grep -E '^queue' | cut -c1-80)"
done
