#!/bin/bash
# Per-dispatch K2 durations of several builds in one GPU session: tools/stats_libs.sh <workload> lib1.so lib2.so ...  ("-" = in-tree)
WL=$1; shift
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$REPO/$lib; fi
  D=/tmp/st_$$_$(basename $lib .so); rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-extra --no-verify --steps 10 --warmup 2 > /dev/null 2>&1
  echo "== $lib"
  python - $D <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
by=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'sfw_social' not in n: continue
    key=('flat' if 'flat' in n else 'reg', r.get('Grid_Size_X') or r.get('Grid_Size'))
    by[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(by.items(), key=lambda kv:-sum(kv[1])/len(kv[1])):
    v=v[len(v)//5:]  # skip warm-up dispatches
    print('   %-5s grid=%-9s n=%-3d mean %.1f us  min %.1f us' % (k[0],k[1],len(v),sum(v)/len(v)/1e3,min(v)/1e3))
P
done
