"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE results db:
    python scratch/r04/mfma_summary.py <pmc_results.db> <out.csv>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
k = collections.defaultdict(dict)
for name, c, v, n in rows: k[name][c] = (v, n)
out = []
for name, d in k.items():
    if 'GRBM_GUI_ACTIVE' not in d: continue
    gui, n = d['GRBM_GUI_ACTIVE']; busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, n))[0]; ins = d.get('SQ_INSTS_MFMA', (0, n))[0]
    simd_cycles = gui / 8.0 * 1024.0            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
    out.append((gui, name.replace('(anonymous namespace)::', '').split('(')[0][:90], n, gui / 8.0 / n, busy / n, ins / n, busy / simd_cycles if simd_cycles else 0.0))
out.sort(reverse=True)
tot_gui = sum(o[0] for o in out); tot_busy = sum(o[4] * o[2] for o in out)
with open(sys.argv[2], 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline\n')
    f.write('# per dispatch; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)\n')
    f.write('# whole run: mfma_util = %.3f\n' % (tot_busy / (tot_gui / 8.0 * 1024.0)))
    f.write('Kernel,Dispatches,CyclesPerDispatch,MfmaBusyCyclesPerDispatch,MfmaInstsPerDispatch,MfmaUtil,ShareOfGpuCycles\n')
    for gui, name, n, cyc, busy, ins, util in out[:40]:
        f.write('"%s",%d,%.0f,%.0f,%.0f,%.3f,%.4f\n' % (name, n, cyc, busy, ins, util, gui / tot_gui))
print(open(sys.argv[2]).read()[:2500])
