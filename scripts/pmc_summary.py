"""Summarise rocprofv3 --pmc passes (one directory per counter set) into profiles/<round>_pmc_summary.json.

usage: pmc_summary.py OUT.json DIR_FETCH DIR_WRITE [DIR_TCC]
Each DIR holds the counter_collection CSV of one `rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv` run of
the same bench command.  Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are reported in KiB-like units of 1024 B here; FETCH_SIZE under-counts wide coalesced reads by 2x (it counts
128-B requests as 64 B), so fetch_bytes_corrected = 2 x raw is the upper estimate used for `roofline.traffic`.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    rows = []
    for p in f:
        rows += list(csv.DictReader(open(p)))
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch (dispatch order)
    for r in sorted(rows, key=lambda r: int(r.get('Dispatch_Id', 0))):
        name = re.sub(r'\(.*', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        per[name][r['Counter_Name']].append(float(r['Counter_Value']))
    return per


def main():
    out, d_fetch, d_write = sys.argv[1:4]
    d_tcc = sys.argv[4] if len(sys.argv) > 4 else None
    F, W = load(d_fetch), load(d_write)
    T = load(d_tcc) if d_tcc else {}
    kernels = {}
    for k in sorted(set(F) | set(W)):
        fv, wv = F.get(k, {}).get('FETCH_SIZE', []), W.get(k, {}).get('WRITE_SIZE', [])
        act = [v for v in fv if v > 0.5]
        e = {'launches': len(fv), 'active_launches': len(act),
             'fetch_kib_mean_active': round(sum(act) / len(act), 1) if act else 0.0, 'fetch_kib_max': round(max(fv), 1) if fv else 0.0}
        wact = [v for v in wv if v > 0.5]
        e['write_kib_mean_active'] = round(sum(wact) / len(wact), 1) if wact else 0.0
        e['write_kib_max'] = round(max(wv), 1) if wv else 0.0
        if k in T:
            h, m = sum(T[k].get('TCC_HIT_sum', [])), sum(T[k].get('TCC_MISS_sum', []))
            e['l2_hit_rate'] = round(h / (h + m), 4) if h + m else None
        kernels[k] = e
    res = {'command': 'rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --repeat 1'
                      '  (one pass per counter set: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)',
           'units': 'FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3; fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 '
                    '(MI355X_MICROARCH.md, HBM section: upper estimate for gather patterns)',
           'kernels': kernels}
    for tag, key in (('k_gn_iter', 'k_gn_iter_full_launch'), ('k_knn5', 'k_knn5_full_search_launch'), ('k_vox_ds_seg', 'k_vox_ds_seg_launch'), ('k_feat_lf_voxel', 'k_feat_lf_voxel_launch'), ('k_odom_corr_grid', 'k_odom_corr_grid_launch'), ('k_vb_reduce', 'k_vb_reduce_launch'), ('k_feat_ring', 'k_feat_ring_launch'),
                     ('k_vb_stack', 'k_vb_stack_launch'), ('k_odom_lm', 'k_odom_lm_launch')):
        kn = [k for k in kernels if tag in k]
        if kn:
            k = kn[0]
            fr = kernels[k]['fetch_kib_max'] * 1024
            wr = kernels[k]['write_kib_max'] * 1024
            res[key] = {'fetch_bytes_raw': int(fr), 'fetch_bytes_corrected': int(2 * fr), 'write_bytes': int(wr), 'traffic_bytes': int(2 * fr + wr)}
            print(key, res[key])
    json.dump(res, open(out, 'w'), indent=1)
    print('wrote', out, 'kernels', len(kernels))


if __name__ == '__main__':
    main()
