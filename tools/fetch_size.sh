# FETCH_SIZE / WRITE_SIZE per access width (tools/micro/fetch_size.hip): bash tools/fetch_size.sh > gpurun_out/fetch_size.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_fs
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_fs -- tools/micro/fetch_size > gpurun_out/pmc_fs.log 2>&1 || { tail -5 gpurun_out/pmc_fs.log; exit 1; }
  python tools/pmc_summary.py gpurun_out/pmc_fs | python -c "
import re, sys
k = None
for line in sys.stdin:
    m = re.match(r'^(\S+)\s+dispatches', line)
    if m: k = m.group(1); continue
    m = re.match(r'^\s+(\w+)\s+([0-9.e+]+)', line)
    if m and k and (('read' in k) == (m.group(1) == 'FETCH_SIZE')):
        kb = float(m.group(2))
        print(f'{k:22s} {m.group(1)} {kb:12.0f} KB for 1048576 KB moved: bytes / counter = {1048576.0 / kb:.3f}')
"
done
rm -rf gpurun_out/pmc_fs
