# kernel-level same-box A/B of the intermediate layout: one-stream graph-replay kernel stats with HNO_MID_ZLAYOUT=1 (default) and =0
bash tools/r5/prof_step.sh r05_g_z1 HNO_MID_ZLAYOUT=1 | grep -E "plain run|dht_|spec_mid|sum of kernel"
bash tools/r5/prof_step.sh r05_g_z0 HNO_MID_ZLAYOUT=0 | grep -E "plain run|dht_|spec_mid|sum of kernel"
