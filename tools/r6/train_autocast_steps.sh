# seconds per training step through training(use_autocast=True), replayed vs eager: the difference of two epoch lengths removes capture / warm-up
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in fnoseg vnet; do
  LAB_NB=24 python3 tools/r6/train_autocast_ab.py $m 2>/dev/null | tail -1 > /tmp/a.json
  LAB_NB=120 python3 tools/r6/train_autocast_ab.py $m 2>/dev/null | tail -1 > /tmp/b.json
  python3 - <<PY
import json
a, b = json.load(open('/tmp/a.json')), json.load(open('/tmp/b.json'))
for mode in ('replayed', 'eager'):
    print('$m', mode, 'ms per training step', round((b[mode]['seconds_total'] - a[mode]['seconds_total']) / (3 * 96) * 1e3, 2), 'losses equal', a['replayed']['train_loss'] == a['eager']['train_loss'])
PY
done
