# usage: [PROBE_ORDERS="shuffle:8 cell:32 ..."] [PROBE_SCENE="3 500"] bash tools/coherence_probe.sh
# extend kernel time on rays that have bounced, by how they are ordered (tools/coherence_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for item in ${PROBE_ORDERS:-shuffle:8 octant:8 cell:8 cell_octant:8 octant_cell:8 cell_octant:16 global:16}; do
  rm -rf /tmp/cp_out
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_out -- python3 $R/tools/coherence_probe.py ${item%%:*} ${item##*:} $PROBE_SCENE 2>&1 | grep -E "^(shuffle|octant|cell|global|asmade)" || exit 1
  python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/cp_out/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_extend' in r['Name']: print('   ', r['Name'][:60], 'calls', r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
PY
done
