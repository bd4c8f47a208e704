"""Per-kernel statistics (count, total / average duration, registers, LDS) from a rocprofv3 rocpd database -> CSV on stdout.
usage: python tools/rocpd_stats.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
                            max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size)
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,ArchVGPR,AccumVGPR,SGPR,LDSBytes,ScratchBytes")
for r in rows:
    name = r[0].replace('"', "'")
    print(f'"{name}",{r[1]},{int(r[2])},{r[3]:.1f},{int(r[4])},{int(r[5])},{100.0 * r[2] / tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}')
