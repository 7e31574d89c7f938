#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> profiles/traffic.json (what bench.py's roofline block reads) and a per-kernel counter summary.

    python3 tools/pmc_traffic.py PASSES_DIR OUT_DIR [TAG]

PASSES_DIR holds one sub-directory per PMC pass, as tools/r03_profiles.sh makes them: `<config>__<passname>/` with rocprofv3's
`*counter_collection.csv` (one row per dispatch and counter) and a `meta.json` {"n": .., "batch": .., "flags": ..}.
Per kernel and counter the value is the MEAN OVER THE LAST 10 LAUNCHES of the pass (steady state: the bench's conditioning and
warm-up launches come first).  Kernel names are folded to their base name (k_td_play_hot<5, 512, false> and <..., true> ->
k_td_play; k_td_play<6, ...> -> k_td_play).

HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) x 1024, the two counters collected in SEPARATE passes (they do not fit one:
MI355X_MICROARCH.md, "rocprofv3 PMC slots"), with the gfx950 correction that guide prescribes: FETCH_SIZE counts a wide
(16 B per lane) coalesced streaming read at HALF its bytes.  The correction is applied to the reads it was calibrated for
only — k_td_play's 16-byte board and 16-byte RNG loads (16 B per lane added); 4-byte gathers, 8-byte index records and the
stores are left as counted (uncalibrated widths, the guide says so).

The output is stamped with the sha256 of 2048_amd/lib2048_hip.so: bench.py refuses the file for any other build.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOLD = [('k_td_play', r'k_td_play(_hot|_lds2|_lds3)?<'), ('k_td_update_owner', r'k_td_update_owner<'), ('k_apply_orbits_mean', r'k_apply_orbits_mean'),
        ('k_apply_orbits', r'k_apply_orbits\('), ('k_hex_count', r'k_hex_count'), ('k_hex_plan', r'k_hex_plan'), ('k_hex_scatter', r'k_hex_scatter'),
        ('k_hex_owner', r'k_hex_owner'), ('k_hex_apply', r'k_hex_apply'), ('k_sort_count', r'k_sort_count'), ('k_sort_scan', r'k_sort_scan'),
        ('k_sort_scatter', r'k_sort_scatter'), ('k_step_random', r'k_step_random'), ('k_eval_select', r'k_eval_select')]


def base_name(kernel):
    for name, pat in FOLD:
        if re.search(pat, kernel):
            return name
    return None


def read_pass(d):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return {}
    series = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            k = base_name(row['Kernel_Name'])
            if k:
                series.setdefault((k, row['Counter_Name']), []).append(float(row['Counter_Value']))
    return {key: sum(v[-10:]) / len(v[-10:]) for key, v in series.items()}


def main():
    passes, out_dir = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else 'r03'
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(ROOT, '2048_amd', 'lib2048_hip.so'), 'rb') as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    configs = {}
    for d in sorted(glob.glob(os.path.join(passes, '*__*'))):
        meta_path = os.path.join(d, 'meta.json')
        if not os.path.exists(meta_path):
            continue
        meta = json.load(open(meta_path))
        key = meta.get('key') or f'n{meta["n"]}_b{meta["batch"]}'
        entry = configs.setdefault(key, {'_meta': meta, '_passes': []})
        entry['_passes'].append(os.path.basename(d))
        for (kernel, counter), value in read_pass(d).items():
            entry.setdefault(kernel, {})[counter] = value
    for key, entry in configs.items():
        lanes = int(entry['_meta']['batch'])
        for kernel, c in entry.items():
            if kernel.startswith('_') or 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
                continue
            wide = 16 * lanes if kernel == 'k_td_play' else 0      # half of the 32 B per lane of 16-byte board + RNG loads
            c['hbm_bytes'] = (c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0 + wide
            c['hbm_read_bytes'] = c['FETCH_SIZE'] * 1024.0 + wide
            c['hbm_write_bytes'] = c['WRITE_SIZE'] * 1024.0
            c['fetch_size_correction_bytes'] = wide
    out = {'so_sha256': sha, 'generated_by': 'tools/pmc_traffic.py (from the rocprofv3 --pmc passes of tools/r03_profiles.sh)',
           'units': 'per launch, mean of the last 10 launches of a pass; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; *_bytes in bytes; '
                    'SQ_* in quad-cycles summed over the chip (MI355X_MICROARCH.md, PMC units)',
           'entries': configs}
    with open(os.path.join(out_dir, 'traffic.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    with open(os.path.join(out_dir, f'{tag}_pmc_summary.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for key, entry in configs.items():
        for kernel, c in sorted(entry.items()):
            if not kernel.startswith('_'):
                print(key, kernel, {k: round(v) for k, v in c.items()})


if __name__ == '__main__':
    main()
