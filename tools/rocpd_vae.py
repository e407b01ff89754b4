import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.workgroup_size_y from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
# last decode: find last 55 conv launches
conv = [r for r in rows if 'gemm_bf16_kernel_v2' in r[0] or 'conv3d_halo' in r[0]]
last = conv[-55:]
t0 = last[0][1]
sel = [r for r in rows if r[1] >= t0]
for name, s, e, gx, wx, gy, wy in sel:
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:48]
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  wgs {gx//max(wx,1):5d} x {gy//max(wy,1):2d}  {n}")
