cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pp; rocprofv3 --kernel-trace -d /tmp/pp -- python $R/tools/nn_microbench.py "$@" > /tmp/log.txt 2>&1
DB=$(find /tmp/pp -name "*_results.db" | head -1)
python - <<PY
import sqlite3
cur=sqlite3.connect("$DB").cursor()
rows=cur.execute("select name, grid_x, grid_y, grid_z, count(*), avg(duration), min(duration) from kernels where name like '%nn_%' group by name, grid_x, grid_y, grid_z order by min(start)").fetchall()
for r in rows: print(r[0][27:75], r[1:4], r[4], round(r[5]/1e3,1), round(r[6]/1e3,1))
PY
