cd $GRAFT_REPO_ROOT
python tools/c3_probe.py 8 none,copy,k1,k2,k4,copy:coherent,copy:noncoherent,k16:noncoherent,k4:noncoherent,k16:coherent > gpurun_out/r4c_modes.json 2> gpurun_out/r4c_modes.err
for e in GPU_FORCE_BLIT_COPY_SIZE=0 DEBUG_CLR_LIMIT_BLIT_WG=2 HSA_FORCE_SDMA_SIZE=1048576 HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE=1 GPU_BLIT_ENGINE_TYPE=1 GPU_BLIT_ENGINE_TYPE=2 ; do
  env $e python tools/c3_probe.py 8 copy,copy:coherent > gpurun_out/r4c_env_$e.json 2> gpurun_out/r4c_env_$e.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k: round(v['chunks_per_s'],1) for k,v in d.items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
