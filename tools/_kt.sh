cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 0; do
rm -rf gpurun_out/kt_$v
timeout -k 5 300 rocprofv3 --kernel-trace -d gpurun_out/kt_$v -o run -- python tools/train_n.py regex1g 31744 fuse_step=$v > gpurun_out/kt_$v.log 2>&1; echo "kt$v rc=$?"
db=$(ls gpurun_out/kt_$v/*/*.db gpurun_out/kt_$v/*.db 2>/dev/null | head -1)
python tools/rocpd_stats.py $db | head -8 | cut -c1-60,200-400
python tools/rocpd_timeline.py $db 60 > gpurun_out/r6_g_timeline_fuse$v.txt
python - $db <<'P'
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
rows=cur.execute("select name,start,end from kernels order by start").fetchall()
rows=rows[-3000:]
import statistics
gaps=[(rows[i][1]-rows[i-1][2])/1e3 for i in range(1,len(rows))]
durs=[(r[2]-r[1])/1e3 for r in rows]
print("last 3000 kernels: median gap %.2f us, mean gap %.2f, median dur %.2f, mean dur %.2f, wall %.1f us per kernel"%(statistics.median(gaps),sum(gaps)/len(gaps),statistics.median(durs),sum(durs)/len(durs),(rows[-1][2]-rows[0][1])/1e3/len(rows)))
P
rm -rf gpurun_out/kt_$v
done
