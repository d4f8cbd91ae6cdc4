for m in heavyhex c1; do
  NREP=10 TNQS_HOST_TIMING=1 python profiles/shape_bench.py $m 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$m', d['ms_per_layer'], {k:(v['ms'],v['launches']) for k,v in d['classes'].items()})
    elif 'host timing' in line: print(line.strip())
"
done
