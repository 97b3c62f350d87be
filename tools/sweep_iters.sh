#!/bin/bash
# bench.py lines (headline + configs) against the wave iterations per tile of the plain scans (KNOX_ITERS) and the workgroups
# per CU (KNOX_BLOCKS_PER_CU), each setting in turn.  usage: [ITERS="2 4"] [PER_CU=2] tools/sweep_iters.sh
for rep in 1 2; do for it in ${ITERS:-2 4}; do
KNOX_BLOCKS_PER_CU=${PER_CU:-2} KNOX_ITERS=$it python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('per_cu=${PER_CU:-2} iters=$it', round(d['value']/1e12,4), round(r['frac'],4), round(r['kernel_ms'],4), round(r['frac_of_read_only_ceiling'],4), [round(x,3) for x in d['placement']['device_quiet_ms']], {k:round(v['frac'],4) for k,v in d['configs'].items()})"
done; done
