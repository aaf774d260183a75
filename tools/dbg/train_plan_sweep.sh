#!/bin/bash
# training plans of the shipped generator specs at batch sizes the census does not reach (it trains at a
# quarter of its inference batch): forward + backward three times, gradients finite?  (tools/dbg/train_plan_probe.py)
cd "$(dirname "$0")/../.."
while read cfg shape; do
  echo -n "$cfg $shape: "
  timeout 300 python tools/dbg/train_plan_probe.py $cfg $shape 2>&1 | grep -v amdgpu | tail -1
done <<'LIST'
spatial/gen_2x_1f.json 48,75,75,1
spatial/gen_2x_2f.json 96,75,75,2
spatial/gen_10x_2f.json 48,20,20,2
spatiotemporal/gen_2x_2x_2f.json 8,20,20,24,2
spatiotemporal/gen_2x_2x_2f.json 16,20,20,6,2
spatiotemporal/gen_2x_12x_14f.json 4,20,20,12,14
spatiotemporal/gen_3x_4x_1f.json 8,20,20,24,1
spatiotemporal/gen_3x_4x_2f.json 12,20,20,5,2
spatiotemporal/gen_3x_4x_10f.json 8,20,20,24,10
spatiotemporal/gen_3x_4x_14f.json 6,22,18,7,14
spatiotemporal/gen_4x_24x_3f.json 4,16,16,8,3
spatiotemporal/gen_4x_24x_3f.json 16,16,16,2,3
sup3rcc/gen_solar_1x_8x_1f.json 12,54,54,3,3
sup3rcc/gen_trh_1x_24x_2f.json 4,54,54,10,4
sup3rcc/gen_trh_1x_24x_2f.json 8,54,54,3,4
sup3rcc/gen_wind_1x_24x_6f.json 2,54,54,10,6
sup3rcc/gen_solar_5x_1x_1f.json 120,16,16,3
sup3rcc/gen_wind_5x_1x_6f.json 120,16,16,7
sup3rcc/gen_wind_3x_4x_2f.json 16,20,20,24,2
LIST
