cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_r04.py -x -q 2>&1 | tail -4
for cfg in "0 0" "1 0" "1 1" "1 2" "1 4" "1 7"; do set -- $cfg
SUP3R_AMD_PERSIST2=$1 SUP3R_AMD_MFMA_DBG=$2 python bench.py --no-train --no-cpu-baseline --no-parity-mode --no-traffic --steps 10 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('persist2=$1 dbg=$2', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
done
