#!/bin/bash
# round 6, call 11: gpslc_predict_multi with direct strided delivery — its tests, then the host-delivery timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c11; mkdir -p $O
timeout -k 10 500 python3 tools/bench_multi.py > $O/multi.json 2> $O/multi.err; echo "bench_multi rc=$?"; tail -2 $O/multi.err
python3 -c "
import json; d=json.load(open('$O/multi.json'))
for k,v in d.items():
    print(k, v['shape']['host_bytes']/1e9, 'GB; compute only', round(v['compute_only_s'],3))
    for name in ('gpslc_predict','gpslc_predict_multi[0]','gpslc_predict_multi[0,0]','round5_staging_restated'):
        if name in v: print('   ', name, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v[name].items()})
"
