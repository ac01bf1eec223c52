cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2; do
HNO_DEFER=0 python3 tools/dbg/vnet_ab.py 0 2>&1 | grep flag | sed 's/^/defer0 /'
HNO_DEFER=1 python3 tools/dbg/vnet_ab.py 0 2>&1 | grep flag | sed 's/^/defer1 /'
done
