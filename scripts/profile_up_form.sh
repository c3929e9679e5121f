cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in cavity_fine refined1; do
for f in column row; do
rm -rf gpurun_out/pc; FC_UP_FORM=$f rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc -- python scripts/bench_case.py $c --steps 60 > /dev/null 2>&1
echo "== $c $f"
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pc/**/*_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# take the timed steps: find the repeating pattern by kernel sequence; here: aggregate by (kernel name, grid) median
import statistics
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if n.startswith('void fc_nd') or n.startswith('fc_nd'):
        d[(n[:48], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=0
for k,v in d.items():
    if len(v)>=50:
        m=statistics.median(v); print(f"{k[0]:50s} grid {k[1]:>9s} calls {len(v):5d} median {m/1000:8.1f} us")
PY
done; done
