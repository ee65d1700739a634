#!/bin/bash
# BASELINE configs[2] THROUGH THE BINDINGS: 10M x 768 written as the reference's table files, loaded by the
# reference's loader (load_db_scaled), queried with query() and query_batch() of the drop-in `epsilla` module.  One gpurun call.
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${ROWS:-10000000}; DIM=${DIM:-768}; NQ=${NQ:-1024}; METRIC=${METRIC:-EUCLIDEAN}; BATCHES=${BATCHES:-10}
TAG=${TAG:-r5_module_10M}
D=/dev/shm/epsdb_$$; mkdir -p $D
(time python $R/scripts/epsilla_module_driver.py $R/dropin/_build $D/db bulk $ROWS $DIM $NQ $METRIC $BATCHES) > $R/gpurun_out/$TAG.log 2>&1
rm -rf $D
grep EPSILLA_JSON $R/gpurun_out/$TAG.log | python -c "
import sys,json
o=json.loads(sys.stdin.read()[len('EPSILLA_JSON '):])
rows=$ROWS
s={'what':'epsilla module (dropin/_build) over libepsilla_gfx950: %d x %d %s, table files written by vectordb_amd/segment_file.py and loaded by the reference loader' % (rows,$DIM,'$METRIC'),
   'generate_s':o['generate_s'],'write_files_s':o['write_files_s'],'load_db_s':o['load_db_s']}
for f in o['single']: s['query() first 16 calls, filter %r, seconds' % f]=o['single'][f]['first16_s']
for f in ('', 'ID < %d' % (rows//2)):
    b=o['batch'][f]
    s['query_batch filter %r' % f]={'qps':b['qps'],'ms_per_batch':b['ms_per_batch'],'queries':b['queries'],'batches':b['batches'],
      'equals_query()_on_first_16':all(o['single'][f]['results'][i][0]==b['results'][i][0] for i in range(16))}
    if 'batch_arrays' in o: s['query_batch(as_arrays=True) filter %r' % f]=o['batch_arrays'][f]
if 'graph' in o:
    s['rebuild() on the device mirror, seconds'] = o['rebuild_s']
    s['query_batch after rebuild (graph traversal, the reference\'s defaults)'] = o['graph']
print(json.dumps(s))
" | tee $R/gpurun_out/$TAG.json
grep -v "^\[20" $R/gpurun_out/$TAG.log | grep -v EPSILLA_JSON | tail -6
