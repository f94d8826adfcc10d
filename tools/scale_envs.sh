#!/bin/bash
# single-GPU scale points and the secondary configurations (run on the GPU box); tools/scale_parse.py writes profiles/r03_scale_envs.json
mkdir -p gpurun_out/scale
run() { tag=$1; shift; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/scale/$tag.log 2>&1 || { echo "$tag failed"; tail -3 gpurun_out/scale/$tag.log; exit 1; }; echo "$tag $(grep -o '"value": [0-9.]*' gpurun_out/scale/$tag.log | head -2 | tr '\n' ' ')"; }
for n in 16384 65536 262144 524288 1048576; do run scale_f_$n --envs $n --steps 500 --warmup 100 || exit 1; done
for n in 4096 16384 262144 524288 1048576; do run scale_c_$n --contacts --envs $n --steps 100 --warmup 50 || exit 1; done
run mixed_free --mixed --envs 12288 || exit 1
run mixed_contacts --mixed --contacts --envs 12288 || exit 1
run c65536_pgs --contacts --envs 65536 --solver pgs --steps 50 --warmup 20 || exit 1
run c65536_cg --contacts --envs 65536 --solver cg --steps 100 --warmup 50 || exit 1
run f4096_cg --envs 4096 --solver cg || exit 1
run f4096_pgs --envs 4096 --solver pgs || exit 1
echo done
