for prio in -1 0; do for order in before after; do
echo "== comm prio $prio order $order"
SRLX_FABRIC_COMM_PRIO=$prio SRLX_FABRIC_ORDER=$order python bench.py --roles-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
l=d['learner_rank']; print('bare %.4f fabric %.4f ratio %.3f | update alone %.4f ingest alone %.4f | actor %.4f'%(l['ms_per_period'],l['fabric_ms_per_period'],l['fabric_over_bare'],l['update_alone_ms'],l['ingest_alone_ms'],d['actor_rank']['ms_per_lock_step']))"
done; done
