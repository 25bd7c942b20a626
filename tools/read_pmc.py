import sqlite3, glob, sys
for db in sorted(glob.glob(sys.argv[1] + '/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    ev = [t for t in tabs if 'pmc_event' in t][0]; info=[t for t in tabs if 'info_pmc' in t][0]; disp=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t]
    q = f"select k.kernel_name, p.name, sum(e.value), count(*) from {ev} e join {info} p on e.pmc_id=p.id join {disp} d on e.event_id=d.event_id join {ks[0]} k on d.kernel_id=k.id group by 1,2"
    for r in con.execute(q):
        if len(sys.argv) < 3 or sys.argv[2] in r[0]: print(r[0][:28], r[1], int(r[2]), "dispatches", r[3])
