"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv) of `bench.py --no-graph` into
profiles/hbm_traffic.json (bytes per launch keyed by the kernel names bench.py reports) and a per-instantiation
table.  usage: make_hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix>

traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the bytes of coalesced streaming
reads (MI355X_MICROARCH.md, HBM section); both counters are in KB."""
import csv, json, sys, collections, re

GROUPS = {
    'pwconv_bwd_kernel': ['pwconv_bwd'], 'pwconv_fwd_kernel': ['pwconv_fwd'], 'dht_fwd_plane_kernel': ['dht_fwd_plane', 'dht_fwd_items'],
    'dht_inv_plane_kernel': ['dht_inv_plane', 'dht_inv_item'], 'spec_mid_fwd_kernel': ['spec_mid_kernel<65, 10, false', 'spec_mid_kernel<33, 10, false'], 'spec_mid_bwd_kernel': ['spec_mid_kernel<65, 10, true', 'spec_mid_kernel<33, 10, true'], 'dht_fwd_d_kernel': ['dht_fwd_d'], 'dht_inv_d_kernel': ['dht_inv_d'],
    'specmix_fwd_kernel': ['specmix_fwd'], 'specmix_bwd_kernel': ['specmix_bwd'], 'reduce_partials_kernel': ['reduce_partials'],
    'conv_k2s2_fwd_kernel': ['conv_k2s2_fwd'], 'conv_k2s2_bwd_kernel': ['conv_k2s2_bwd', 'conv_k2s2_chain_bwd'], 'upsoftmax_fwd_kernel': ['upsoftmax_fwd', 'uphead_seg', 'uphead_rows'],
    'upsoftmax_bwd_kernel': ['upsoftmax_bwd_plane', 'upsoftmax_bwd_kernel'], 'upsoftmax_bwd_d_kernel': ['upsoftmax_bwd_d'],
    'loss_stats_kernel': ['loss_stats'], 'loss_bwd_kernel': ['loss_bwd'], 'chan_restride_kernel': ['chan_restride'],
}


def read(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        name = re.sub(r'^void ', '', r['Kernel_Name'])
        name = re.sub(r'^hno::', '', name).split('(')[0]
        a = acc[name]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


def main():
    fetch, write, prefix = read(sys.argv[1], 'FETCH_SIZE'), read(sys.argv[2], 'WRITE_SIZE'), sys.argv[3]
    per_inst = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(('at::', '__amd', 'Cijk')):
            continue
        n = fetch.get(k, write.get(k))[0]
        per_inst[k] = {'launches': n, 'FETCH_SIZE_KB_per_launch': round(fetch.get(k, (0, 0.0))[1], 1),
                       'WRITE_SIZE_KB_per_launch': round(write.get(k, (0, 0.0))[1], 1)}
    out = {}
    for g, pats in GROUPS.items():
        n = fe = wr = 0.0
        for k, v in per_inst.items():
            if any(k.startswith(p) for p in pats):
                n += v['launches']
                fe += v['launches'] * v['FETCH_SIZE_KB_per_launch']
                wr += v['launches'] * v['WRITE_SIZE_KB_per_launch']
        if n:
            out[g] = round((2.0 * fe + wr) * 1024 / n)
    json.dump(out, open('profiles/hbm_traffic.json', 'w'), indent=1)
    json.dump({'note': __doc__, 'per_kernel_bytes_per_launch': out, 'per_instantiation': per_inst}, open(prefix + '_pmc_hbm_traffic.json', 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
