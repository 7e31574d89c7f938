#!/usr/bin/env python3
"""Fold the per-pass JSON files of tools/pmc_many.sh into profiles/rNN_pmc_summary.json and refresh profiles/traffic.json.

    python3 tools/pmc_summarise.py r02 gpurun_out/r02_profiles/pmc "lane sort on (default)" gpurun_out/r02_profiles/pmc_nosort "G2048_SORT_EVERY=0"

Values are per launch, mean of the last 10 launches of a pass.  traffic.json (read by bench.py for roofline.traffic) is
built from the FIRST directory's FETCH_SIZE / WRITE_SIZE passes with the correction the microarchitecture guide prescribes
for wide streaming reads, applied to the reads it was calibrated for only (k_td_play's 16-byte board and RNG loads)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANES = 1 << 20


def fold(d):
    out = {}
    for p in sorted(glob.glob(os.path.join(d, 'pmc_*.json'))):
        try:
            j = json.load(open(p))
        except ValueError:
            continue
        for kern, ctrs in j.items():
            for name, v in ctrs.items():
                out.setdefault(kern, {})[name] = v['mean_last10']
    return out


def main():
    tag, rest = sys.argv[1], sys.argv[2:]
    summary = {}
    first = None
    for d, label in zip(rest[0::2], rest[1::2]):
        summary[label] = fold(d)
        first = first or summary[label]
    summary['_comment'] = ('rocprofv3 --pmc passes (two TA/TCP counters per pass) of `python3 bench.py --steps 20 --warmup 20 --repeats 1`; '
                           'per launch, mean of the last 10 launches; 2^20 lanes, n = 5.  FETCH_SIZE / WRITE_SIZE in KB.')
    json.dump(summary, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_summary.json'), 'w'), indent=1)
    wide = LANES * 32 // 2          # k_td_play reads 16 B of board + 16 B of RNG per lane in 16-byte loads: counted at half
    traffic = {'_comment': 'HBM-side bytes per launch, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes '
                           f'(profiles/{tag}_pmc_summary.json), 2^20 lanes, n = 5, lane sort on. Correction per MI355X_MICROARCH.md (HBM): '
                           'FETCH_SIZE counts a wide (16 B/lane) coalesced streaming read at half its bytes - applied to k_td_play\'s board '
                           'and RNG reads only (33.5 MB actual, 16.8 MB added); the 4-byte table gathers and the 8+4-byte record scans are '
                           'uncalibrated widths and are left as counted. Far below the algorithmic bytes (428 B x 2^20 = 449 MB for '
                           'k_td_play): the table is L2 / Infinity-Cache resident.',
               'raw': {}}
    for kern, key in (('k_td_play', 'k_td_play<5>'), ('k_td_update_owner<5>', 'k_td_update_owner<5>'), ('k_apply_orbits', 'k_apply_orbits')):
        c = first.get(kern, {})
        if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
            continue
        b = (c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 + (wide if kern == 'k_td_play' else 0)
        traffic[f'{key}_b{LANES}'] = int(round(b))
        traffic['raw'][key] = {'FETCH_SIZE_KB': c['FETCH_SIZE'], 'WRITE_SIZE_KB': c['WRITE_SIZE']}
    json.dump(traffic, open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith('_') and k != 'raw'}))


if __name__ == '__main__':
    main()
