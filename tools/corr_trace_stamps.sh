cd /tmp && export TMPDIR=/tmp
for l in 13 12 26; do
rm -rf /tmp/kt$l
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$l -o kt -- python $GRAFT_REPO_ROOT/tools/block_timeline.py --layer $l 2>&1 | grep -E "RAW|event-timed|wall clock|segment medians"
python - <<PY
import csv,glob
f=glob.glob('/tmp/kt$l/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the stamped run is the 4th run_batch (3 warmups + 1), profile run is the 5th; print the kernels of the 4th run around layer $l
conv=[r for r in rows if 'tf2::' in r['Kernel_Name']]
per=len(conv)//5
run=conv[3*per:4*per]
for i,r in enumerate(run):
    if abs(i-($l+1))<=1 or i<2:
        print(i, r['Kernel_Name'][:50], r['Start_Timestamp'], r['End_Timestamp'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us; gap to next start', (int(run[i+1]['Start_Timestamp'])-int(r['End_Timestamp']))/1e3 if i+1<len(run) else None)
PY
done
