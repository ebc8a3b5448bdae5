"""Summarise rocprofv3 outputs (kernel trace + --pmc passes) for one kernel into a small JSON.
usage: python tools/pmc_summary.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]
Every DIR is the -d directory of one rocprofv3 pass (…/<host>/<pid>_{counter_collection,kernel_trace}.csv).  Counter
values and durations are averaged over the dispatches of kernels whose name contains KERNEL_SUBSTRING (the first two
dispatches are dropped as warm-up)."""
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
out_path, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {'kernel_filter': needle, 'passes': dirs, 'counters': {}}
for d in dirs:
    for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if needle in row['Kernel_Name']:
                per.setdefault(row['Counter_Name'], {}).setdefault(row['Dispatch_Id'], 0.0)
                per[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
                res['vgpr'] = int(row['VGPR_Count']); res['lds_bytes'] = int(row['LDS_Block_Size'])
                res['grid'] = int(row['Grid_Size']); res['kernel'] = row['Kernel_Name'][:160]
        for name, disp in per.items():
            vals = [v for _, v in sorted(disp.items(), key=lambda kv: int(kv[0]))][2:] or list(disp.values())
            res['counters'][name] = sum(vals) / len(vals)
    for f in glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True):
        durs = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if needle in r['Kernel_Name']]
        if durs:
            durs = durs[2:] or durs
            res.setdefault('duration_us_by_pass', {})[os.path.basename(d.rstrip('/'))] = sum(durs) / len(durs) / 1e3
c = res['counters']
dur = res.get('duration_us_by_pass', {})
if dur:
    res['duration_us'] = min(dur.values())
if 'GRBM_GUI_ACTIVE' in c and dur:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back); use the duration of ITS pass
    res['effective_clock_GHz'] = c['GRBM_GUI_ACTIVE'] / 8 / (max(dur.values()) * 1e3)
if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
    # busy cycles summed over 1024 SIMDs; chip cycles = GRBM_GUI_ACTIVE / 8
    res['mfma_busy_fraction'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8)
if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
    res['lds_conflict_ratio'] = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']
if 'SQ_WAIT_ANY' in c and c.get('SQ_WAVE_CYCLES'):
    res['wave_wait_fraction'] = c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']
if 'SQ_WAIT_INST_ANY' in c and c.get('SQ_WAVE_CYCLES'):
    res['wave_issue_stall_fraction'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
if 'FETCH_SIZE' in c:
    res['hbm_read_bytes'] = c['FETCH_SIZE'] * 1024 * 2      # KB units; gfx950 reports half of a wide streaming read
if 'WRITE_SIZE' in c:
    res['hbm_write_bytes'] = c['WRITE_SIZE'] * 1024
json.dump(res, open(out_path, 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ('passes',)}, indent=1))
