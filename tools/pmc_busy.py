#!/usr/bin/env python
"""Which unit a kernel keeps busy, from a tools/pmc_train.sh / pmc_bench.sh
summary: VALU / MFMA / LDS busy fractions and resident waves per SIMD.

    python tools/pmc_busy.py profiles/r04/pmc_train.txt [min_kilocycles]

Units (MI355X_MICROARCH.md, rocprofv3 counter notes): GRBM_GUI_ACTIVE is
summed over the 8 XCDs; SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU and
SQ_ACTIVE_INST_LDS count quad-cycles (x 4) summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1 024 SIMDs."""
import re
import sys

SIMDS = 1024


def main():
    txt = open(sys.argv[1]).read()
    min_kc = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    rows = []
    for blk in re.split(r'\n(?=\S)', txt):
        lines = blk.strip().split('\n')
        d = {}
        for ln in lines[1:]:
            m = re.match(r'\s+(\S+)\s+mean\s+([\d.]+)\s+n (\d+)', ln)
            if m:
                d[m.group(1)] = float(m.group(2))
                d['n'] = int(m.group(3))
        if 'GRBM_GUI_ACTIVE' not in d:
            continue
        cyc = d['GRBM_GUI_ACTIVE'] / 8
        if cyc < min_kc * 1e3:
            continue
        rows.append((cyc * d['n'], lines[0].strip(), cyc,
                     d.get('SQ_ACTIVE_INST_VALU', 0) * 4 / SIMDS / cyc,
                     d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / SIMDS / cyc,
                     d.get('SQ_ACTIVE_INST_LDS', 0) * 4 / SIMDS / cyc,
                     d.get('SQ_WAVE_CYCLES', 0) * 4 / SIMDS / cyc, d['n']))
    print('| kernel | launches | kilocycles | VALU busy | MFMA busy | LDS busy | waves / SIMD |')
    print('|---|---|---|---|---|---|---|')
    for _, name, cyc, valu, mfma, lds, occ, n in sorted(rows, reverse=True):
        print(f'| `{name}` | {n} | {cyc / 1e3:.0f} | {valu:.2f} | {mfma:.2f} | {lds:.2f} | {occ:.1f} |')


if __name__ == '__main__':
    main()
