cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in 8 4; do echo "waves $w"; HNO_PWCHAIN_WAVES=$w python3 tools/dbg/pw_chain_lab.py 2>&1 | tail -2; done
