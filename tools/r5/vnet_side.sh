cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "HNO_WGRAD_SIDE=$v"; HNO_WGRAD_SIDE=$v python3 tools/dbg/vnet_ab.py 0 2>&1 | grep "ms per step"; done
