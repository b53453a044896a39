cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=$PWD
for rep in 1 2; do
for lib in "" "$PWD/align_anything_amd/libaa_hip_old.so"; do
AA_HIP_LIB=$lib timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-gemm-events > gpurun_out/b.json 2> gpurun_out/b.err
python -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('lib=${lib##*/}', d['value'], d['ms_per_step'], d['step_mfma']['frac_of_dense_bf16_peak'])"
done; done
