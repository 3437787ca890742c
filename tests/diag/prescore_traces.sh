cd /root/repo
for m in 125m 350m; do for r in 16 64; do for p in "" "--prescore"; do
python bench.py --trace gamma --trace-rate $r --model $m $p 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=j['trace']
f=lambda d: '%.3f/%.3f/%.3f' % (d['p50'],d['p95'],d['p99'])
print('$m rate $r $p:', 'with_arrivals', f(t['ranker_ms_with_arrivals']), 'incl hooks', f(t['ranker_ms_with_arrivals_incl_hooks']), 'steady p50 %.3f' % t['ranker_ms_steady']['p50'], 'share_hol %.4f' % t['ranker_share_of_hol'], 'hol p50 %.2f' % t['hol_ms']['p50'], 'pre', j['ranker_metrics']['prescore']['launches'], j['ranker_metrics']['prescore']['requests'])
"
done; done; done
