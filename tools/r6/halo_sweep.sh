# sweep of the halo kernel tile shapes (HNO_HALO_SHAPE=MT,NT,KS) on the level-2 / level-3 V-Net layers: bash tools/r6/halo_sweep.sh
# (round 6, last run: the plan picked by cb_conv_impl is within 3-7 % of the best forced shape on every layer: l3_192_192 29.2 / 25.3 us against 28.3 / 24.8 for <2,1,8>; l2_96_96 31.5 / 27.2 against 29.4 / 25.3 for <2,1,4>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for sh in l3_192_192 l2_96_96 l2_192_96; do
for cfg in default 1,1,8 1,1,4 1,1,2 1,2,8 1,2,4 1,3,8 1,3,4 2,1,8 2,1,4 2,2,4 2,3,4 1,3,2; do
  if [ $cfg = default ]; then unset HNO_HALO_SHAPE; else export HNO_HALO_SHAPE=$cfg; fi
  echo -n "$sh $cfg: "; python3 tools/bench_cb_conv.py $sh 2>/dev/null | tail -1 | cut -c1-110
done; done
