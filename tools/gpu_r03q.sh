set -x
mkdir -p gpurun_out/r03q
python tools/bench_ops.py --only gemm,tn,norm,misc --variants -1 --vendor 2>&1 | grep -v amdgpu > gpurun_out/r03q/bench_ops_auto_vendor.txt
cat gpurun_out/r03q/bench_ops_auto_vendor.txt
