cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/dbg/vnet_ab.py $FLAGS 2>&1 | grep flag
