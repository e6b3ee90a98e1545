V=$PWD/ray-tracing-practice_amd/variants
cd /tmp && export TMPDIR=/tmp
for v in default ntl nts; do
  if [ $v = default ]; then unset RTP_AMD_LIB; else export RTP_AMD_LIB=$V/librtp_amd_$v.so; fi
  echo "== $v"; SPP=500 ITERS=2 python3 $GRAFT_REPO_ROOT/tools/perf_sweep.py | grep -o "best kernel ms [0-9.]*\|trace ms [0-9.]*"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$v_$c; SPP=500 ITERS=1 rocprofv3 --pmc $c --output-format csv -d /tmp/pm_${v}_$c -o pmc -- python3 $GRAFT_REPO_ROOT/tools/perf_sweep.py > /dev/null 2>&1
    python3 - /tmp/pm_${v}_$c $c <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float);cnt=collections.Counter()
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]; acc[k]+=float(r['Counter_Value']); cnt[k]+=1
for k in acc:
    if 'render_kernel<true, false' in k or 'primary' in k or 'accumulate_kernel<false' in k: print('   ',sys.argv[2],k[:50],'%.2f GB per launch'%(acc[k]/cnt[k]*1024/1e9*(2 if sys.argv[2]=='FETCH_SIZE' else 1)))
PY
  done
done
