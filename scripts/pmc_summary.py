import csv, glob, collections, sys, os
d = sys.argv[1]
allc = {}
main_p1 = None
for f in sorted(glob.glob(os.path.join(d, 'p*_counter_collection.csv'))):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
        by[int(r['Dispatch_Id'])]['_grid'] = int(r['Grid_Size'])
    if not by:
        print(f, 'EMPTY'); continue
    # the dominant launch = the LONGEST coarse dispatch of this pass' own kernel trace (the main stage of a full batch; the side
    # measurements of the bench also launch coarse kernels - small batches - so "the last dispatch" is not it)
    pick = max(by)
    tr = f.replace('_counter_collection.csv', '_kernel_trace.csv')
    if os.path.exists(tr):
        rows_t = [r for r in csv.DictReader(open(tr)) if 'coarse' in r.get('Kernel_Name', '') and int(r['Dispatch_Id']) in by]
        if rows_t:
            best = max(rows_t, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            pick = int(best['Dispatch_Id'])
            if os.path.basename(f).startswith('p1_'):
                main_p1 = (best['Kernel_Name'], int(best['End_Timestamp']) - int(best['Start_Timestamp']))
    last = by[pick]
    for k, v in last.items():
        allc[k] = v
for k in sorted(allc):
    print(f"{k:34s} {allc[k]:16.1f}")
g = allc.get
if g('SQ_WAVE_CYCLES'):
    wc = g('SQ_WAVE_CYCLES')
    print("-- fractions of SQ_WAVE_CYCLES: WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f" % (
        g('SQ_WAIT_ANY', 0) / wc, g('SQ_WAIT_INST_ANY', 0) / wc, g('SQ_ACTIVE_INST_ANY', 0) / wc))
if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('GRBM_GUI_ACTIVE'):
    # GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs
    cyc = g('GRBM_GUI_ACTIVE') / 8
    print("-- kernel cycles/XCD %.0f ; MFMA busy per SIMD %.0f ; MFMA util %.3f" % (
        cyc, g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024, g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc))
if g('SQ_LDS_IDX_ACTIVE'):
    print("-- LDS bank conflict fraction %.3f" % (g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE')))
if g('FETCH_SIZE'):
    print("-- FETCH_SIZE KB %.0f (x2 gfx950 correction => %.2f GB), WRITE_SIZE KB %.0f" % (
        g('FETCH_SIZE'), g('FETCH_SIZE') * 2 * 1024 / 1e9, g('WRITE_SIZE', 0)))
if g('TCC_REQ_sum'):
    print("-- TCC hit rate %.3f" % (g('TCC_HIT_sum', 0) / (g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 1))))

# duration of the same dispatch in the pass that carried GRBM_GUI_ACTIVE (the PMC passes run with --kernel-trace): the clock the
# launch ran at = cycles per XCD / duration (say what limits the launch - the nominal peaks assume 2.4 GHz)
dur_ns = main_p1[1] if main_p1 else None
if main_p1:
    print("-- dominant launch:", main_p1[0][:90])
if dur_ns and g('GRBM_GUI_ACTIVE'):
    clk = g('GRBM_GUI_ACTIVE') / 8 / dur_ns
    print("-- profiled duration %.1f us ; clock %.3f GHz (GRBM_GUI_ACTIVE / 8 / duration)" % (dur_ns / 1e3, clk))
    allc['profiled_duration_us'] = dur_ns / 1e3
    allc['clock_ghz'] = clk
    if g('SQ_VALU_MFMA_BUSY_CYCLES'):
        allc['mfma_busy'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (g('GRBM_GUI_ACTIVE') / 8)

import json
out = {k: allc[k] for k in allc if not k.startswith('_')}
if g('FETCH_SIZE'):
    # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced
    # reads by exactly 2x (MI355X_MICROARCH.md, HBM section) -> corrected here
    out['hbm_bytes_per_launch'] = (g('FETCH_SIZE') * 2 + g('WRITE_SIZE', 0)) * 1024
# stamp: the kernel sources this pass ran on (bench.py quotes `traffic` only from a file whose stamp matches its own)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out['source_sha16'] = bench.source_sha16()
except Exception as e:  # noqa: BLE001
    out['source_sha16'] = None
    print('-- no source stamp:', e)
json.dump(out, open(os.path.join(d, 'pmc_main_kernel.json'), 'w'), indent=1, sort_keys=True)
