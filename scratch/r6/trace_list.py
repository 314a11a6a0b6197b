"""List the dispatches of a rocprofv3 --kernel-trace database in order: name, grid (workgroups), block, LDS, VGPRs, duration.
    python scratch/r6/trace_list.py <db> [skip-substring ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
q = ("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.workgroup_size_y, d.group_segment_size, s.arch_vgpr_count, d.end - d.start "
     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start")
for k, gx, gy, gz, wx, wy, lds, vg, ns in c.execute(q):
    if any(x in k for x in sys.argv[2:]): continue
    wgs = (gx // max(wx, 1)) * (gy // max(wy, 1)) * max(gz, 1)
    print('%-70s wgs %7d x %4d thr  lds %6d vgpr %3d  %8.1f us' % (k.replace('_ZN12_GLOBAL__N_1', '')[:70], wgs, wx * wy, lds, vg, ns / 1e3))
