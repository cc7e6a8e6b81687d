"""Add the records of a find-db recording (tools/record_find_db.sh -> gpurun_out/fdb_record) that the package does not ship yet to
creamfl_amd/miopen_db (text find-db / perf-db: one `key=value` line per problem) and creamfl_amd/miopen_cache (sqlite: compiled
kernels).  Records the package already ships are kept VERBATIM (the per-call find-db gate answers for them because they were
measured against the timed search).      python tools/merge_find_db.py [gpurun_out/fdb_record]"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'fdb_record')
added = {}
for fn in sorted(os.listdir(os.path.join(src, 'db'))):
    if not fn.endswith('.txt'):
        continue
    dst = os.path.join(ROOT, 'creamfl_amd', 'miopen_db', fn)
    have = {}
    if os.path.exists(dst):
        for ln in open(dst):
            if '=' in ln:
                have[ln.split('=', 1)[0]] = ln
    new = [ln if ln.endswith('\n') else ln + '\n' for ln in open(os.path.join(src, 'db', fn)) if '=' in ln and ln.split('=', 1)[0] not in have]
    if new:
        with open(dst, 'a') as f:
            f.writelines(new)
    added[fn] = len(new)
for fn in sorted(os.listdir(os.path.join(src, 'cache'))):
    if not fn.endswith('.ukdb'):
        continue
    dst = os.path.join(ROOT, 'creamfl_amd', 'miopen_cache', fn)
    con = sqlite3.connect(dst)
    con.execute("attach database ? as rec", (os.path.join(src, 'cache', fn),))
    cols = [r[1] for r in con.execute('pragma table_info(kern_db)') if r[1] != 'id']
    before = con.execute('select count(*) from kern_db').fetchone()[0]
    con.execute('insert into kern_db (%s) select %s from rec.kern_db r where not exists (select 1 from kern_db k where k.kernel_name = '
                'r.kernel_name and k.kernel_args = r.kernel_args)' % (','.join(cols), ','.join('r.' + c for c in cols)))
    con.commit()
    added[fn] = con.execute('select count(*) from kern_db').fetchone()[0] - before
    con.execute('detach database rec')
    con.execute('pragma journal_mode=delete')
    con.execute('vacuum')
    con.close()
print(added)
