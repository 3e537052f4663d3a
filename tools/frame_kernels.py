#!/usr/bin/env python3
"""Per-dispatch durations of the last frame in a rocprofv3 rocpd database (gpurun_out/<dir>/f_results.db)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = c.execute("select name, (end-start)/1e3, grid_x, workgroup_x, vgpr_count, scratch_size from kernels where name like '%rtmi%' order by start").fetchall()
for r in rows[-last:]:
    print("%-44s %9.1f us grid %8d wg %4d vgpr %3d scratch %d" % (r[0][:44], r[1], r[2], r[3], r[4], r[5]))
