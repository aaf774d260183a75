"""HBM bytes per launch of the training step's HBM-class kernels (C2, batch 8)
from a tools/pmc_train.sh summary, next to their algorithmic bytes.

    python tools/pmc_table.py profiles/r03/pmc_train.txt [kernel_stats.txt]

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream at 64 B (MI355X_MICROARCH.md, HBM): the
table shows it raw and doubled; TCC_MISS x 128 B (read AND write misses of the
L2) is the cross-check that tells which of the two applies to a kernel."""
import re
import sys

P0 = 8 * 80 * 80 * 288          # hi-res positions
P1 = 8 * 78 * 78 * 286          # after the first (valid) discriminator conv
P2 = 8 * 38 * 38 * 142          # after the stride-2 conv
GB = 1e9
# kernel -> (what, algorithmic read bytes, algorithmic write bytes)
ALG = {
    'gconv_fewch_halo_kernel<2, 1, 2>': ('disc 2->32 forward', P0 * 8, P1 * 64),
    # (round 4: the kernel gained its split-bf16 template flag)
    'gconv_fewch_halo_kernel<2, 1, 2, false>': ('disc 2->32 forward', P0 * 8, P1 * 64),
    'conv_halo_s2_kernel<2>': ('disc 32->32 s2 forward', P1 * 64, P2 * 64),
    'conv_dgrad_s2_kernel<2, true, true>': ('disc 32->32 s2 data gradient (fp32 dPre + sign bytes in, bf16 out)',
                                            P2 * 128 + P1 * 4, P1 * 64),
    'conv_dgrad_c2_slide_kernel': ('disc 2->32 data gradient', P1 * 64, P0 * 8),
    'conv_wgrad_c2_kernel<2, 2, true, false>': ('disc 2->32 weight gradient', P1 * 64 + P0 * 8, 0),
    'conv_wgrad_bf16_gen_kernel<2, 2, true, true, false>': ('disc 32->32 s2 weight gradient', P1 * 64 + P2 * 128, 0),
    'conv_tail_slide_kernel': ('generator 8->2 tail forward', P0 * 16, P0 * 8),
    'conv_wgrad_tail_kernel<true>': ('generator 8->2 weight gradient', P0 * 16 + P0 * 8, 0),
    # round 6: the plane-sweep forms
    'conv_tail_sweep_kernel<3>': ('generator 8->2 tail forward (plane sweep)', P0 * 16, P0 * 8),
    'conv_wgrad_tail_sweep_kernel': ('generator 8->2 weight gradient (plane sweep)', P0 * 16 + P0 * 8, 0),
}


def main():
    txt = open(sys.argv[1]).read()
    avg = {}
    if len(sys.argv) > 2:
        for line in open(sys.argv[2]):
            m = re.match(r'(\S.*?)\s+calls\s+\d+\s+total\s+[\d.]+ ms\s+avg\s+([\d.]+) us', line)
            if m:
                avg[m.group(1).strip()] = float(m.group(2))
    print('| kernel | layer | algorithmic read + write (GB) | FETCH raw / x2 (GB) | WRITE (GB) | '
          'L2 misses x 128 B (GB) | fetch / algorithmic read | avg (us) | (read + write) / time |')
    print('|---|---|---|---|---|---|---|---|---|')
    for blk in re.split(r'\n(?=\S)', txt):
        lines = blk.strip().split('\n')
        name = lines[0].strip()
        if name not in ALG:
            continue
        d = {}
        for ln in lines[1:]:
            m = re.match(r'\s+(\S+)\s+mean\s+([\d.]+)', ln)
            if m:
                d[m.group(1)] = float(m.group(2))
        what, rd, wr = ALG[name]
        raw = d.get('FETCH_SIZE', 0) * 1e3 / GB
        wsz = d.get('WRITE_SIZE', 0) * 1e3 / GB
        miss = d.get('TCC_MISS_sum', 0) * 128 / GB
        # the doubled figure applies when it (plus the writes) explains the L2 misses
        x2 = abs(2 * raw + wsz - miss) < abs(raw + wsz - miss)
        fetch = 2 * raw if x2 else raw
        us = avg.get(name)
        rate = '%.2f TB/s = %.2f' % ((rd + wr) / (us * 1e-6) / 1e12, (rd + wr) / (us * 1e-6) / 8e12) if us else ''
        print('| `%s` | %s | %.3f + %.3f | %.3f / %.3f%s | %.3f | %.3f | %.2f | %s | %s |' % (
            name, what, rd / GB, wr / GB, raw, 2 * raw, ' (x2 applies)' if x2 else ' (raw applies)', wsz, miss,
            fetch / (rd / GB), ('%.0f' % us) if us else '', rate))


if __name__ == '__main__':
    main()
