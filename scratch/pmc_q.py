import sqlite3,sys
for db in sys.argv[1:]:
    c=sqlite3.connect(db)
    q=("select s.kernel_name, count(*), sum(p.value), avg(d.end-d.start) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name")
    for r in c.execute(q): print(db.split('/')[-2], r[0][:60], r[1], 'KiB/launch %.0f'%(r[2]/r[1]), 'ns %.0f'%r[3])
