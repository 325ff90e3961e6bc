#!/usr/bin/env python3
"""Summarise the three rocprofv3 passes of tools/prof_corr.py (kernel trace / FETCH_SIZE / WRITE_SIZE) for one shape:
    python tools/prof_corr_summary.py <dir_trace> <dir_fetch> <dir_write> B C H W dtype  ->  <out>.txt + <out>_pmc.json
HBM traffic per launch as MI355X_MICROARCH.md prescribes: separate --pmc passes, FETCH_SIZE in KiB units x2 (gfx950
reports half of the bytes of wide coalesced reads; calibrated in the same pass on the bf16-cast kernel of known size),
WRITE_SIZE in KiB."""
import csv
import glob
import json
import os
import sys


def rows(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def main():
    d_tr, d_f, d_w = sys.argv[1:4]
    B, C, H, W = (int(v) for v in sys.argv[4:8])
    dtype = sys.argv[8]
    out = sys.argv[9]
    s = 2 if dtype != 'fp32' else 4
    alg_r, alg_w = s * B * H * W * 2 * C, s * B * H * W * 81
    tr = sorted(rows(d_tr, 'kernel_trace.csv'), key=lambda r: int(r['Start_Timestamp']))
    corr = [r for r in tr if 'corr81' in r['Kernel_Name']]
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in corr]
    n = len(dur) // 2
    cold, warm = dur[:n], dur[n:]
    name = corr[0]['Kernel_Name'].split('(')[0]

    def pmc(d, counter):
        rr = [r for r in rows(d, 'counter_collection.csv') if r['Counter_Name'] == counter]
        k = [float(r['Counter_Value']) for r in rr if 'corr81' in r['Kernel_Name']]
        cal = [float(r['Counter_Value']) for r in rr if 'bfloat16_copy' in r['Kernel_Name'] or 'copy_kernel' in r['Kernel_Name']]
        return k, cal
    fk, fcal = pmc(d_f, 'FETCH_SIZE')
    wk, wcal = pmc(d_w, 'WRITE_SIZE')
    nf = len(fk) // 2
    fetch_cold, fetch_warm = sum(fk[:nf]) / nf, sum(fk[nf:]) / max(1, len(fk) - nf)
    nw = len(wk) // 2
    write_cold = sum(wk[:nw]) / nw
    fetch_bytes = fetch_cold * 1024 * 2
    write_bytes = write_cold * 1024
    # corr81_allc_kernel<T, UW, NU, NT, RAGGED, NORM, TPW, OC8, PADW, TO>
    targs = [a.strip() for a in name[name.index('<') + 1:name.rindex('>')].split(',')] if '<' in name else []
    norm_ = len(targs) > 5 and targs[5] == 'true'
    oc8_ = len(targs) > 7 and targs[7] == 'true'
    variant = 'norm_c8' if (norm_ and oc8_) else 'norm' if norm_ else 'plain'
    summ = {'shape': [B, C, H, W], 'dtype': dtype, 'kernel': name, 'variant': variant, 'padw': len(targs) > 8 and targs[8] == 'true', 'algorithmic_bytes': alg_r + alg_w, 'algorithmic_read_bytes': alg_r,
            'algorithmic_write_bytes': alg_w, 'fetch_bytes_corrected_x2': fetch_bytes, 'write_bytes': write_bytes,
            'traffic_bytes': fetch_bytes + write_bytes, 'traffic_over_algorithmic': (fetch_bytes + write_bytes) / (alg_r + alg_w),
            'calibration_cast_kernel_fetch_KB': fcal[:2], 'calibration_cast_kernel_bytes_read': 4 * B * C * H * W,
            'kernel_us_cold_avg': sum(cold) / len(cold), 'kernel_us_cold_min': min(cold),
            'kernel_us_warm_avg': sum(warm) / len(warm), 'kernel_us_min': min(warm),
            'frac_of_8TBs_warm': (alg_r + alg_w) / (sum(warm) / len(warm)) / 8e6, 'frac_of_8TBs_cold': (alg_r + alg_w) / (sum(cold) / len(cold)) / 8e6}
    json.dump({'FETCH_SIZE': {'per_launch_KB_cold': fetch_cold, 'per_launch_KB_warm': fetch_warm, 'calibration_copy_kernel_KB': fcal[:2]},
               'WRITE_SIZE': {'per_launch_KB_cold': write_cold, 'calibration_copy_kernel_KB': wcal[:2]}, 'summary': summ},
              open(out + '_pmc.json', 'w'), indent=1)
    with open(out + '_rocprof.txt', 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/prof_corr.py %d %d %d %d %s   (MI355X, corr81 forward)\n' % (B, C, H, W, dtype))
        f.write('# %d launches with the 256 MiB infinity cache evicted in between + %d back-to-back launches\n' % (len(cold), len(warm)))
        agg = {}
        for r in tr:
            k = r['Kernel_Name'][:120]
            d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            a = agg.setdefault(k, [0, 0, 1 << 60, 0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        f.write('%-122s %6s %12s %10s %8s %8s\n' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns'))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('%-122s %6d %12d %10d %8d %8d\n' % (k, a[0], a[1], a[1] // a[0], a[2], a[3]))
        f.write('\n%s per-dispatch duration (us): cache-evicted launches avg %.2f min %.2f | back-to-back launches avg %.2f min %.2f\n'
                % (name, summ['kernel_us_cold_avg'], summ['kernel_us_cold_min'], summ['kernel_us_warm_avg'], summ['kernel_us_min']))
        f.write('algorithmic bytes s*B*H*W*(2C+81) = %d  ->  %.1f %% of 8 TB/s back-to-back, %.1f %% cache-evicted\n'
                % (alg_r + alg_w, 100 * summ['frac_of_8TBs_warm'], 100 * summ['frac_of_8TBs_cold']))
        f.write('\nPMC (separate passes): FETCH_SIZE %.1f KB/launch, WRITE_SIZE %.1f KB/launch (cache-evicted launches)\n' % (fetch_cold, write_cold))
        f.write('  corrected read bytes (FETCH_SIZE x 1024 x 2, gfx950 correction) = %.2f MB vs algorithmic read %.2f MB\n' % (fetch_bytes / 1e6, alg_r / 1e6))
        f.write('  write bytes (WRITE_SIZE x 1024) = %.2f MB vs algorithmic write %.2f MB\n' % (write_bytes / 1e6, alg_w / 1e6))
        f.write('  traffic / algorithmic = %.3f\n' % summ['traffic_over_algorithmic'])
        if fcal:
            f.write('  same pass: the fp32->%s cast kernel reads %d B and reports FETCH_SIZE %.0f KB = %.2f of its bytes (the x2 correction)\n'
                    % (dtype, 4 * B * C * H * W, fcal[0], fcal[0] * 1024 / (4 * B * C * H * W)))
    print(open(out + '_rocprof.txt').read())


if __name__ == '__main__':
    main()
