#!/bin/bash
# rocprofv3 counter passes over an arbitrary command (each group its own run, --kernel-trace only):
#   bash tools/pmc_cmd.sh <outdir> '<kernel-name regex>' -- <command ...>
# writes <outdir>/pmc_summary.txt: mean per dispatch of every counter for the kernels matching the regex
OUT=$1; PAT=$2; shift; shift; shift
ROOTD=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOTD/$OUT"; CMD=("$@")
for i in "${!CMD[@]}"; do [[ -e "$ROOTD/${CMD[$i]}" ]] && CMD[$i]="$ROOTD/${CMD[$i]}"; done
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT" \
 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$ROOTD/$OUT/pass$i" -- "${CMD[@]}" > "$ROOTD/$OUT/pass$i.log" 2>&1
done
python3 - "$ROOTD/$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    return k.split('(')[0].replace('void ', '')[:80]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r['Kernel_Name']):
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fo:
    for k, d in sorted(agg.items()):
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            fo.write(f'   {c:34s} mean/dispatch {sum(v)/len(v):18.1f}  n={len(v)}\n')
print(open(out + '/pmc_summary.txt').read())
PY
rm -rf "$ROOTD/$OUT"/pass*/
