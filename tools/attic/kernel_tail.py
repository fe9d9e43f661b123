import sqlite3, glob, sys
db=glob.glob(sys.argv[1]+'/**/*.db', recursive=True)[0]
con=sqlite3.connect(db); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t][0]
cols=[r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol='kernel_name' if 'kernel_name' in cols else cols[-1]
rows=list(cur.execute(f"select d.start, d.end, s.{namecol}, d.stream_id, d.workgroup_size_x, d.grid_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t_end=rows[-1][1]
sel=[r for r in rows if r[0] > t_end-30e6]
t0=sel[0][0]
for r in sel:
    print("%8.3f %8.3f  %-48s stream %s wg %d grid %d" % ((r[0]-t0)/1e6, (r[1]-r[0])/1e6, r[2][:48], r[3], r[4], r[5]))
