cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 2 4 6 8 10 12 16; do echo -n "nsplit $n: "; HNO_HM_NSPLIT=$n python3 tools/dbg/hmha_one.py 2>&1 | tail -1; done
