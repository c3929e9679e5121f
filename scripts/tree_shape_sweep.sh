for cfg in "10 2" "8 2" "9 3" "12 2" "12 3" "10 1" "11 2"; do
  set -- $cfg
  FC_ND_DEPTH=$1 FC_ND_MERGE=$2 timeout -k 10 200 python bench.py --steps 1000 --no-cpu-baseline --no-large-spmv 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('depth $1 merge $2:', round(d['value']), 'steps/s; sweeps', round(d['phase_ms_eager']['sweeps']*1e3,1), 'us;', r['launches_per_step'], 'launches;', round(r['bytes_per_launch']*r['launches_per_step']/1e6,1), 'MB; factor_nnz', r['factor_nnz'])"
done
